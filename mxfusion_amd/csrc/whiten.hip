// Pieces of the WHITENED float32 tier of the SVGP training call (composite.hip; svgp_regression.py:83-92 solves with the Cholesky factor --
// this tier is that factorised form on the split GEMMs):
//
//     V = L^-1 Kuf          (split GEMM, triangular A, written directly as f16x2 planes in the (m, k = n) orientation: gemm_split.hip c_blk = 2)
//     Phi = V V^T           (split GEMM on those planes: the statistic the core's reverse mode consumes, d/dC = P beta / 2 Phi)
//     T = L^-T (I - A_s A_s^T) V = Hh V     (split GEMM: needs V as the (n, k = m) operand -> the transposition below)
//     U = a^T V, a = L^-1 mu                (fused into the transposition pass for P = 1)
//
// The explicit-inverse tier forms T = H0 Kuf with |H0| ~ cond(Kuu): its float32 rounding error grows like cond 2^-24.  Here every operand
// is bounded by |L^-1| ~ sqrt(cond) and |v_n|^2 <= k_nn, so the error grows like sqrt(cond) 2^-24.
#include "common.h"
#include "internal.h"

namespace {

typedef unsigned int w_u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------- planes transposition
// in : two f16 planes of an (R x K) operand, element (r, k) at ((k / 16) * R + r) * 16 + k % 16       (R = M rows, K = SB columns)
// out: the same values as the (K x R) operand, element (k, r) at ((r / 16) * K + k) * 16 + r % 16
// One workgroup = 64 consecutive k (four 16-blocks) x ALL rows, walked in chunks of 64 rows through an LDS tile.  Reads and writes are
// whole 2 KB runs (64 rows x 32 bytes / 64 k x 32 bytes).  HBM bound: 4 bytes read + 4 bytes written per element.
// PT = 1: the same pass forms U[k] = scale * sum_r a[r] (hi + lo)(r, k)   (the row a^T V of the whitened tier; scale = sigma 2^-14).
constexpr int TRS = 72;                 // LDS row stride in 16-bit elements (144 bytes: 16-byte aligned rows)
template <int PT>
__global__ __launch_bounds__(256) void planes_transpose_kernel(int64_t R, int64_t K, const unsigned short* __restrict__ in, int64_t pin,
                                                               unsigned short* __restrict__ out, int64_t pout, const float* __restrict__ a,
                                                               const float* __restrict__ scale, float sc2, float* __restrict__ U) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[2][64 * TRS];
    __shared__ float usum[64];
    const int t = threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.x * 64;                  // first k of this workgroup (K % 64 == 0: host)
    if (PT > 0 && t < 64) usum[t] = 0.f;
    float uacc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) uacc[i][j] = 0.f;
    for (int64_t r0 = 0; r0 < R; r0 += 64) {
        // ---- read 64 rows x 64 k: unit u = t + 256 i  ->  k block u >> 7, row (u & 127) >> 1, k half u & 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = t + 256 * i, kb = u >> 7, rl = (u & 127) >> 1, half = u & 1;
            const int64_t off = ((k0 / 16 + kb) * R + r0 + rl) * 16 + half * 8;
            const w_u32x4 vh = *reinterpret_cast<const w_u32x4*>(in + off);
            const w_u32x4 vl = *reinterpret_cast<const w_u32x4*>(in + pin + off);
            const float ar = PT > 0 ? a[r0 + rl] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned short h = (unsigned short)((j & 1) ? (vh[j >> 1] >> 16) : (vh[j >> 1] & 0xffffu));
                const unsigned short l = (unsigned short)((j & 1) ? (vl[j >> 1] >> 16) : (vl[j >> 1] & 0xffffu));
                const int kl = kb * 16 + half * 8 + j;
                tile[0][kl * TRS + rl] = h;
                tile[1][kl * TRS + rl] = l;
                if (PT > 0) uacc[i][j] = fmaf(ar, (float)__builtin_bit_cast(_Float16, h) + (float)__builtin_bit_cast(_Float16, l), uacc[i][j]);
            }
        }
        __syncthreads();
        // ---- write 64 k x 64 rows: unit u -> row block u >> 7, k (u & 127) >> 1, row half u & 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = t + 256 * i, rb = u >> 7, kl = (u & 127) >> 1, half = u & 1;
            const w_u32x4 vh = *reinterpret_cast<const w_u32x4*>(&tile[0][kl * TRS + rb * 16 + half * 8]);
            const w_u32x4 vl = *reinterpret_cast<const w_u32x4*>(&tile[1][kl * TRS + rb * 16 + half * 8]);
            const int64_t off = ((r0 / 16 + rb) * K + k0 + kl) * 16 + half * 8;
            __builtin_nontemporal_store(vh, reinterpret_cast<w_u32x4*>(out + off));
            __builtin_nontemporal_store(vl, reinterpret_cast<w_u32x4*>(out + pout + off));
        }
        __syncthreads();
    }
    if (PT > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = t + 256 * i, kb = u >> 7, half = u & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(&usum[kb * 16 + half * 8 + j], uacc[i][j]);
        }
        __syncthreads();
        if (t < 64) U[k0 + t] = usum[t] * scale[0] * sc2;
    }
}

// U[n] = scale * sc2 * sum_p Upart[p][n]: the 128-row partial sums the planes-output product's epilogue leaves (gemm_split.hip, avec / Upart),
// added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void upart_reduce_kernel(int64_t N, int nparts, const float* __restrict__ Upart, const float* __restrict__ scale,
                                                           float sc2, float* __restrict__ U) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += Upart[(int64_t)p * N + n];
    U[n] = s * (scale ? scale[0] : 1.f) * sc2;
}

// dst = tril(src) (float64, n x n): potrf leaves the strict upper triangle of its buffer as it was
__global__ void tril_copy_kernel(int64_t n, const double* __restrict__ src, double* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (i % n <= i / n) ? src[i] : 0.0;
}

}  // namespace

int mxf_planes_transpose_internal(mxf_ctx* h, int64_t R, int64_t K, const unsigned short* in, int64_t pin, unsigned short* out, int64_t pout,
                                  const float* a, const float* scale, float sc2, float* U, hipStream_t st) {
    if (R <= 0 || K <= 0) return 0;
    if ((R % 64) != 0 || (K % 64) != 0) MXF_FAIL(h, -2, "planes transpose: R and K must be multiples of 64 (%lld x %lld)", (long long)R, (long long)K);
    if (K / 64 > 2147483647LL) MXF_FAIL(h, -3, "planes transpose: grid too large");
    if (U) hipLaunchKernelGGL(planes_transpose_kernel<1>, dim3((unsigned)(K / 64)), dim3(256), 0, st, R, K, in, pin, out, pout, a, scale, sc2, U);
    else hipLaunchKernelGGL(planes_transpose_kernel<0>, dim3((unsigned)(K / 64)), dim3(256), 0, st, R, K, in, pin, out, pout, a, scale, sc2, U);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

int mxf_upart_reduce_internal(mxf_ctx* h, int64_t N, int nparts, const float* Upart, const float* scale, float sc2, float* U, hipStream_t st) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(upart_reduce_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, nparts, Upart, scale, sc2, U);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

int mxf_tril_copy_internal(mxf_ctx* h, int64_t n, const double* src, double* dst, hipStream_t st) {
    int64_t b = (n * n + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(tril_copy_kernel, dim3((unsigned)b), dim3(256), 0, st, n, src, dst);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// ---- C ABI: the two operations for callers that chain split products (include/mxf_gp.h) ---------------------------------------------
extern "C" int mxf_gemm_f16x2_planes_out(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                                         const void* B_planes, const void* B_maxword, void* C_planes, void* Ct_planes, const void* a, void* U,
                                         int a_lower, void* stream) {
    if (!h) return -1;
    if (M <= 0 || N <= 0 || K <= 0 || !A_planes || !B_planes || !A_maxword || !B_maxword || !C_planes) MXF_FAIL(h, -2, "mxf_gemm_f16x2_planes_out: bad argument");
    if ((a || U) && !(a && U && Ct_planes)) MXF_FAIL(h, -2, "mxf_gemm_f16x2_planes_out: U needs a, U and Ct_planes");
    float* upart = nullptr;
    if (U) {
        upart = (float*)mxf_ws(h, sizeof(float) * (size_t)(M / 128) * (size_t)N);
        if (!upart) MXF_FAIL(h, -4, "mxf_gemm_f16x2_planes_out: cannot allocate the partial sums");
    }
    int rc = mxf_gemm_split_internal(h, M, N, K, alpha, (const unsigned short*)A_planes, (int64_t)mxf_split_plane_elems(M, K), (const unsigned short*)B_planes,
                                     (int64_t)mxf_split_plane_elems(N, K), 0.0, nullptr, N, 0, (hipStream_t)stream, 0, MXF_SPLIT_F16X2, nullptr, 0,
                                     (const unsigned*)A_maxword, (const unsigned*)B_maxword, 0, nullptr, (unsigned short*)C_planes,
                                     (int64_t)mxf_split_plane_elems(M, N), a_lower, (unsigned short*)Ct_planes, (int64_t)mxf_split_plane_elems(N, M),
                                     (const float*)a, upart);
    if (rc || !U) return rc;
    return mxf_upart_reduce_internal(h, N, (int)(M / 128), upart, nullptr, 1.f, (float*)U, (hipStream_t)stream);
}

extern "C" int mxf_f16x2_planes_transpose(mxf_handle h, int64_t R, int64_t K, const void* planes_in, void* planes_out, const void* a,
                                          const void* scale, void* U, void* stream) {
    if (!h) return -1;
    if (!planes_in || !planes_out || (U && (!a || !scale))) MXF_FAIL(h, -2, "mxf_f16x2_planes_transpose: bad argument");
    return mxf_planes_transpose_internal(h, R, K, (const unsigned short*)planes_in, (int64_t)mxf_split_plane_elems(R, K), (unsigned short*)planes_out,
                                         (int64_t)mxf_split_plane_elems(K, R), (const float*)a, (const float*)scale, 1.f, (float*)U, (hipStream_t)stream);
}

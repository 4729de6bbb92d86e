// Fused composites: one C-ABI call = one reference `compute` body (value + reverse mode), gfx950.
//
//   mxf_gp_logpdf   <- GPRegressionLogPdf.compute        (modules/gp_modules/gp_regression.py:42-76)
//   mxf_svgp_logpdf <- SVGPRegressionLogPdf.compute      (modules/gp_modules/svgp_regression.py:43-109)
//
// The reference gets gradients from MXNet autograd through potrf/trsm; here the reverse mode is closed
// form (matrix-calculus identities with K^-1 from the Cholesky factor), so every O(n^3) piece is an MFMA
// GEMM and no reverse-mode Cholesky is needed.
//
// SVGP is evaluated in the streaming sufficient-statistics form.  With Ki = Kuu^-1,
// H0 = Ki - Ki Su Ki,  w = Ki mu,  k_n = Kuf[:, n],  beta = 1/noise:
//     l_s   = -B P/2 (log 2pi + log noise) - P beta B var/2 - beta/2 sum_n |y_n - w^T k_n|^2
//             + P beta/2 sum_n k_n^T H0 k_n
//     logL_s = scaling * l_s + negKL,
//     negKL = P/2 (M + logdet Su - logdet Kuu - tr(Ki Su)) - 1/2 tr(mu^T Ki mu)
// All samples' columns are laid side by side (Kuf_all : M x (S*B)), so the data term is TWO big MFMA
// GEMMs ( [H0; w^T] * Kuf_all and Kuf_all * Kuf_all^T ) instead of S batched trsm/gemm2 pairs, and the
// (M x M) factorisation work is done ONCE (not S times: runtime_variable.py:96-99 broadcast) in float64.
#include "common.h"
#include "internal.h"

namespace {

constexpr double LOG2PI = 1.8378770664093453;

// ------------------------------------------------------------------------------------ small kernels
template <typename TI, typename TO>
__global__ void convert_kernel(int64_t rows, int64_t cols, const TI* __restrict__ src, int64_t lds_, TO* __restrict__ dst, int64_t ldd) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * ldd + c] = (TO)src[r * lds_ + c];
    }
}
// float64 -> float32 copy of a contiguous array AND the bit pattern of max |dst| (out: cleared by the caller; non-negative floats order like
// unsigned integers) in one launch -- the M x M operands of the split products went through a convert and a max|x| launch on the chain's critical path
__global__ __launch_bounds__(256) void convert_max_kernel(int64_t n, const double* __restrict__ src, float* __restrict__ dst, unsigned* __restrict__ out) {
    __shared__ unsigned wm[4];
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (float)src[i];
        dst[i] = v;
        const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = wm[w] > m ? wm[w] : m;
        atomicMax(out, m);
    }
}
// dst[c][r] = src[r][c]
template <typename TI, typename TO>
__global__ void transpose_convert_kernel(int64_t rows, int64_t cols, const TI* __restrict__ src, int64_t lds_, TO* __restrict__ dst, int64_t ldd) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        dst[c * ldd + r] = (TO)src[r * lds_ + c];
    }
}
// dst = a*x + b*y (elementwise, contiguous)
template <typename T>
__global__ void axpby_kernel(int64_t n, T a, const T* __restrict__ x, T b, const T* __restrict__ y, T* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = a * x[i] + (y ? b * y[i] : (T)0);
}
// the gradient outputs of the SVGP training call's tail in ONE launch: dst_k (+)= (T)(src_k + src2_k), k < cnt  (six dependent 5-us launches
// at the end of the caller's stream otherwise)
struct FinishArgs { const double* src[6]; const double* src2[6]; void* dst[6]; int64_t n[6]; int acc[6]; int cnt; };
template <typename T>
__global__ void svgp_finish_kernel(FinishArgs a) {
    const int k = blockIdx.y;
    if (k >= a.cnt) return;
    T* __restrict__ d = (T*)a.dst[k];
    const double* __restrict__ s1 = a.src[k];
    const double* __restrict__ s2 = a.src2[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = s1[i] + (s2 ? s2[i] : 0.0);
        d[i] = (a.acc[k] ? d[i] : (T)0) + (T)v;
    }
}
// dst(TO) (+)= a * (TO)src(TI)
template <typename TI, typename TO>
__global__ void add_convert_kernel(int64_t n, TO a, const TI* __restrict__ src, TO* __restrict__ dst, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (accumulate ? dst[i] : (TO)0) + a * (TO)src[i];
}
template <typename T>
__global__ void diag_embed_kernel(int64_t n, const T* __restrict__ d, T* __restrict__ A) {   // A = diag(d)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += (int64_t)gridDim.x * blockDim.x)
        A[i] = (i / n == i % n) ? d[i / n] : (T)0;
}
template <typename TI, typename TO>
__global__ void diag_extract_kernel(int64_t n, const TI* __restrict__ A, int64_t lda, TO* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (TO)A[i * lda + i];
}
// out (+)= scale * sum_i x[i]*y[i]
template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(int64_t n, const T* __restrict__ x, const T* __restrict__ y, double scale, double* __restrict__ out) {
    __shared__ double red[16];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (double)x[i] * (double)y[i];
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) atomic_add(out, s * scale);
}
// out += sum_ij A[i][j] * B[j][i]   (n x n, float64; tr(A B))
__global__ __launch_bounds__(256) void dot_t_kernel(int64_t n, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ out) {
    __shared__ double red[16];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += (int64_t)gridDim.x * blockDim.x) s += A[i] * B[(i % n) * n + i / n];
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) atomic_add(out, s);
}
// copy lower triangle to upper (tiled transpose through LDS)
template <typename T>
__global__ __launch_bounds__(256) void symmetrize_kernel(T* __restrict__ A, int64_t n, int64_t lda, int64_t sA) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;   // tile (by, bx) of the LOWER part, bx <= by
    if (bx > by) return;
    T* a = A + (int64_t)blockIdx.z * sA;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = (int64_t)by * 32 + r, col = (int64_t)bx * 32 + tx;
        tile[r][tx] = (row < n && col < n) ? a[row * lda + col] : (T)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = (int64_t)bx * 32 + r, col = (int64_t)by * 32 + tx;   // transposed position
        if (row < n && col < n && col > row) a[row * lda + col] = tile[tx][r];
    }
}
// The same with a rank-P term folded in (r06, exact GP):  A <- sym(lower(A)) + c v v^T,  v (n x P) -- the reverse mode's 1/2 alpha alpha^T
// without a pass of its own over the n x n matrix
template <typename T>
__global__ __launch_bounds__(256) void symmetrize_rankp_kernel(T* __restrict__ A, int64_t n, int64_t lda, const T* __restrict__ v, int P, T c, int mirror = 1) {
    __shared__ T tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = (int64_t)by * 32 + r, col = (int64_t)bx * 32 + tx;
        T x = 0;
        if (row < n && col < n && col <= row) {
            T d = 0;
            for (int p = 0; p < P; ++p) d = fma(v[row * P + p], v[col * P + p], d);
            x = fma(c, d, A[row * lda + col]);
            A[row * lda + col] = x;
        }
        tile[r][tx] = x;
    }
    if (!mirror) return;         // (block-uniform: the reverse pass reads the lower triangle only)
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t row = (int64_t)bx * 32 + r, col = (int64_t)by * 32 + tx;
        if (row < n && col < n && col > row) A[row * lda + col] = tile[tx][r];
    }
}
// Skinny products with the lower-triangular L^-1 (r06, exact GP; P <= 8 right-hand sides): the general small-product kernel took 0.20 / 0.18 ms
// for these two at n = 8192 -- they read n^2 / 2 doubles, 0.07 ms at the rate a streaming read reaches.
//   trmv_lower_kernel:    y = A x      one wave per row, the lanes along it
//   trmv_lower_t_kernel:  y += A^T x   one thread per column over a chunk of 128 rows, chunks summed with atomics (y zeroed by the caller)
template <typename T>
__global__ __launch_bounds__(256) void trmv_lower_kernel(int64_t n, int P, const T* __restrict__ A, int64_t lda, const T* __restrict__ x, int64_t ldx, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    T acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0;
    const T* a = A + r * lda;
#pragma unroll 4
    for (int64_t c = lane; c <= r; c += 64) {
        const T v = a[c];
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p < P) acc[p] = fma(v, x[c * ldx + p], acc[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < P) { const T t = wave_sum(acc[p]); if (lane == 0) y[r * P + p] = t; }
    }
}
// y = A x for a full (n x n) matrix, one wave per row (r06: w = Ki mu of the SVGP chain -- the small-product kernel with its split-K pre-scale took 0.2 ms
// of the few-sample step's critical path for an 8 MB read)
template <typename T>
__global__ __launch_bounds__(256) void gemv_rows_kernel(int64_t n, int P, const T* __restrict__ A, int64_t lda, const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                        T c = 0, const T* __restrict__ z = nullptr /* y = A x + c z */) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    T acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0;
    const T* a = A + r * lda;
#pragma unroll 4
    for (int64_t c = lane; c < n; c += 64) {
        const T v = a[c];
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p < P) acc[p] = fma(v, x[c * ldx + p], acc[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < P) { const T t = wave_sum(acc[p]); if (lane == 0) y[r * P + p] = z ? t + c * z[r * P + p] : t; }
    }
}
// y (n x P) = A (n x K, row-major, long rows) x (K x P): one workgroup per row (r06: R = Kuf Eb of the generic path, 512 x 131 072 -- the general product took
// 0.18 ms for a 268 MB read); float64 accumulation of the workgroup's partial sums
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(int64_t K, int P, const T* __restrict__ A, int64_t lda, const T* __restrict__ x, T* __restrict__ y) {
    __shared__ double red[16];
    const T* a = A + (int64_t)blockIdx.x * lda;
    T acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0;
#pragma unroll 4
    for (int64_t c = threadIdx.x; c < K; c += 256) {
        const T v = a[c];
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p < P) acc[p] = fma(v, x[c * P + p], acc[p]);
    }
    for (int p = 0; p < P; ++p) {
        const double t = block_sum<double>((double)acc[p], red);
        if (threadIdx.x == 0) y[(int64_t)blockIdx.x * P + p] = (T)t;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void trmv_lower_t_kernel(int64_t n, int P, const T* __restrict__ A, int64_t lda, const T* __restrict__ x, T* __restrict__ y) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x, r0 = (int64_t)blockIdx.y * 128;
    const int64_t r1 = r0 + 128 < n ? r0 + 128 : n;
    if (r1 <= (int64_t)blockIdx.x * 256) return;                 // the chunk lies above the diagonal
    __shared__ T xs[128 * 8];
    for (int e = threadIdx.x; e < (int)(r1 - r0) * P; e += 256) xs[e] = x[r0 * P + e];
    __syncthreads();
    if (c >= n) return;
    T acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0;
    const int64_t rs = c > r0 ? c : r0;
#pragma unroll 4
    for (int64_t r = rs; r < r1; ++r) {
        const T v = A[r * lda + c];
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p < P) acc[p] = fma(v, xs[(r - r0) * P + p], acc[p]);
    }
    if (rs < r1) {
#pragma unroll
        for (int p = 0; p < 8; ++p) if (p < P) atomic_add(y + c * P + p, acc[p]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(int64_t n, const T* __restrict__ x, int64_t sx, double* __restrict__ out) {
    __shared__ double red[16];
    const T* p = x + (int64_t)blockIdx.x * sx;
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)p[i] * (double)p[i];
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void trace_kernel(int64_t n, const T* __restrict__ A, int64_t lda, int64_t sA, T* __restrict__ out) {
    __shared__ double red[16];
    const T* a = A + (int64_t)blockIdx.x * sA;
    double s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)a[i * lda + i];
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (T)s;
}
template <typename T>
__global__ void gp_finalize_kernel(int S, int64_t N, int P, const T* __restrict__ sld, const double* __restrict__ ss, T* __restrict__ logL) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) logL[s] = (T)(-(double)P * (double)sld[s] - 0.5 * (ss[s] + (double)N * P * LOG2PI));
}
// Y (S|1,n) -> dst (S,n), optionally negated
template <typename T>
__global__ void bcast_copy_kernel(int S, int64_t n, const T* __restrict__ src, int64_t ssrc, T* __restrict__ dst, T scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)S * n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = scale * src[(i / n) * ssrc + (i % n)];
}

// dot_kernel ends with ONE atomic per workgroup on one word: 4096 of them serialise to ~0.16 ms (seen on the critical chain in front of the T
// product, r03 timeline); 64 workgroups with a grid-stride loop take ~5 us
inline unsigned dotgrid(int64_t n) { int64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 64) b = 64; return (unsigned)b; }
inline unsigned gridn(int64_t n) { int64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (unsigned)b; }

struct Carver {   // bump allocator over the handle's scratch
    char* base; size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename U> U* take(size_t n) { U* p = (U*)(base + off); off += mxf_align(n * sizeof(U)); return p; }
};

// ================================================================================================ exact GP
template <typename T>
int gp_logpdf_typed(mxf_ctx* h, int kind, int dtype, int S, int64_t N, int Q, int P, const T* X, int64_t sX, const T* Y, int64_t sY,
                    const T* noise, int64_t snoise, const T* ls, int ard, int64_t sls, const T* var, int64_t svar, double jitter,
                    T* logL, T* L, T* LinvY, int* info, int want_grad, T* dX, T* dY, T* dnoise, T* dls, T* dvar, hipStream_t st) {
    const int64_t NN = N * N, NP = N * P;
    size_t need = mxf_align(S * sizeof(T)) + mxf_align(S * sizeof(double));
    if (want_grad) need += 2 * mxf_align((size_t)S * NN * sizeof(T)) + mxf_align((size_t)S * NP * sizeof(T));
    void* ws = mxf_ws(h, need);
    if (!ws) MXF_FAIL(h, -4, "mxf_gp_logpdf: cannot allocate %zu bytes of scratch", need);
    Carver cv(ws);
    T* sld = cv.take<T>(S);
    double* ss = cv.take<double>(S);
    int rc;
    // K = k(X,X) + (noise + jitter) I   (gp_regression.py:55-60), built straight into the L buffer
    rc = mxf_gram(h, kind, dtype, S, N, N, Q, X, sX, nullptr, 0, ls, ard, sls, var, svar, noise, snoise, jitter, MXF_WRITE, L, N, NN, st);
    if (rc) return rc;
    // L^-1 Y (:66).  Large N with gradients: the reverse mode needs L^-1 anyway, and L^-1 Y as ONE product replaces N / 64 dependent
    // block steps (6.9 ms at N = 8192); small N keeps the reference's trsm.
    T* Linv = nullptr;
    const bool via_inverse = want_grad && N >= 2048;
    if (via_inverse) Linv = cv.take<T>((size_t)S * NN);
    // (r06) one matrix, few right-hand sides: the two skinny products with L^-1 as triangular streaming reads, 1/2 alpha alpha^T folded into
    // the symmetrisation of dK (MXF_GP_TRI_SKINNY, probe builds)
    static const int tri_skinny_env = (int)MXF_KNOB("MXF_GP_TRI_SKINNY", 1);
    const bool tri_skinny = via_inverse && S == 1 && P <= 8 && tri_skinny_env != 0;
    // (r05) float64, one matrix: the factorisation forms L^-1 itself, row block by row block on a third stream next to its serial chain
    // (r06) ... and -P/2 L^-T L^-1, the bulk of dlogL/dK, is accumulated row block by row block of that inverse as well (the chip is half idle
    // under the factorisation's serial chain; behind it only the last row block's share is left)
    bool inv_done = false, kacc_done = false;
    T* dK_early = (tri_skinny && sizeof(T) == 8) ? cv.take<T>((size_t)S * NN) : nullptr;
    rc = mxf_potrf_internal(h, dtype, S, N, L, N, NN, info, st, true, true, (via_inverse && S == 1 && sizeof(T) == 8) ? (void*)Linv : nullptr, N, &inv_done,
                            dK_early, N, -0.5 * P, &kacc_done);   // :61
    if (rc) return rc;
    if (via_inverse) {
        if (!inv_done) rc = mxf_trtri_internal(h, dtype, S, N, L, N, NN, Linv, N, NN, st);
        if (rc) return rc;
        if (tri_skinny) hipLaunchKernelGGL((trmv_lower_kernel<T>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, N, P, (const T*)Linv, N, Y, (int64_t)P, LinvY);
        else {
        rc = mxf_gemm_internal(h, dtype, 0, 0, N, P, N, 1.0, Linv, N, NN, Y, P, sY, 0.0, LinvY, P, NP, S, 0, st);
        if (rc) return rc;
        }
    } else {
        hipLaunchKernelGGL((bcast_copy_kernel<T>), dim3(gridn(S * NP)), dim3(256), 0, st, S, NP, Y, sY, LinvY, (T)1);
        rc = mxf_trsm_internal(h, dtype, 0, S, N, P, L, N, NN, LinvY, P, NP, 0, st);
        if (rc) return rc;
    }
    rc = mxf_sumlogdiag_internal(h, dtype, S, N, L, N, NN, sld, st);                    // :67
    if (rc) return rc;
    hipLaunchKernelGGL((sumsq_kernel<T>), dim3(S), dim3(256), 0, st, NP, LinvY, NP, ss);
    hipLaunchKernelGGL((gp_finalize_kernel<T>), dim3((S + 63) / 64), dim3(64), 0, st, S, N, P, sld, ss, logL);   // :68-70
    MXF_LAUNCH_CHECK(h);
    if (!want_grad) return 0;

    // reverse mode: dlogL/dK = 1/2 (alpha alpha^T - P K^-1), alpha = K^-1 Y; dlogL/dY = -alpha
    if (!via_inverse) {
        Linv = cv.take<T>((size_t)S * NN);
        rc = mxf_trtri_internal(h, dtype, S, N, L, N, NN, Linv, N, NN, st);
        if (rc) return rc;
    }
    T* dK = dK_early ? dK_early : cv.take<T>((size_t)S * NN);
    // (r06, probe knob, off) one matrix, Q <= 16: dK stays a LOWER triangle (no mirror pass) and the Gram's reverse pass walks the lower pairs only, twice
    // each, with both sides.  Correct (exact-GP tests pass with it), not faster: MAP step 13.71-13.75 ms without, 13.74-13.77 with -- half the pairs, but the
    // row side is back and the early row bands' blocks leave the chip half empty.
    static const int lower_bwd_env = (int)MXF_KNOB("MXF_GP_LOWER_BWD", 0);
    const bool lower_bwd = lower_bwd_env && tri_skinny && Q <= 16;
    T* alpha = cv.take<T>((size_t)S * NP);
    if (tri_skinny) {
        MXF_HIP(h, hipMemsetAsync(alpha, 0, sizeof(T) * NP, st));
        hipLaunchKernelGGL((trmv_lower_t_kernel<T>), dim3((unsigned)((N + 255) / 256), (unsigned)((N + 127) / 128)), dim3(256), 0, st, N, P, (const T*)Linv, N,
                           (const T*)LinvY, alpha);                                                                 // alpha = Linv^T LinvY
        if (!kacc_done) {
        rc = mxf_gemm_internal(h, dtype, 1, 0, N, N, N, -0.5 * P, Linv, N, NN, Linv, N, NN, 0.0, dK, N, NN, S, 1, st, 0, 1);
        if (rc) return rc;
        }
        hipLaunchKernelGGL((symmetrize_rankp_kernel<T>), dim3((unsigned)((N + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, st, dK, N, N, (const T*)alpha, P, (T)0.5,
                           lower_bwd ? 0 : 1);
    } else {
    rc = mxf_gemm_internal(h, dtype, 1, 0, N, P, N, 1.0, Linv, N, NN, LinvY, P, NP, 0.0, alpha, P, NP, S, 0, st);   // alpha = Linv^T LinvY
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 0, 1, N, N, P, 0.5, alpha, P, NP, alpha, P, NP, 0.0, dK, N, NN, S, 1, st);     // 1/2 alpha alpha^T (lower)
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 1, 0, N, N, N, -0.5 * P, Linv, N, NN, Linv, N, NN, 1.0, dK, N, NN, S, 1, st, 0, 1);  // - P/2 Linv^T Linv (lower; L^-1 lower triangular: tile (i, j <= i) needs k >= i only)
    if (rc) return rc;
    hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)((N + 31) / 32), (unsigned)((N + 31) / 32), S), dim3(256), 0, st, dK, N, N, NN);
    }
    if (dY) hipLaunchKernelGGL((bcast_copy_kernel<T>), dim3(gridn(S * NP)), dim3(256), 0, st, S, NP, (const T*)alpha, NP, dY, (T)-1);
    if (dnoise) hipLaunchKernelGGL((trace_kernel<T>), dim3(S), dim3(256), 0, st, N, (const T*)dK, N, NN, dnoise);
    const int lsn = ard ? Q : 1;
    if (dX) MXF_HIP(h, hipMemsetAsync(dX, 0, sizeof(T) * S * N * Q, st));
    if (dls) MXF_HIP(h, hipMemsetAsync(dls, 0, sizeof(T) * S * lsn, st));
    if (dvar) MXF_HIP(h, hipMemsetAsync(dvar, 0, sizeof(T) * S, st));
    // outputs are per sample (S, ...) even when the primal is broadcast: use per-sample strides for the outputs
    // by running one sample at a time when the primal stride is 0
    for (int s = 0; s < S; ++s) {
        rc = mxf_gram_bwd_internal(h, kind, dtype, 1, N, N, Q, X + (int64_t)s * sX, 0, nullptr, 0, ls + (int64_t)s * sls, ard, 0,
                                   var + (int64_t)s * svar, 0, dK + (int64_t)s * NN, N, NN,
                                   dX ? dX + (int64_t)s * N * Q : nullptr, nullptr, dls ? dls + (int64_t)s * lsn : nullptr,
                                   dvar ? dvar + s : nullptr, st, lower_bwd ? 2 : 1 /* dK was symmetrised above, or only its lower triangle is read */);
        if (rc) return rc;
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// ================================================================================================ SVGP
// per column n (of all S*B columns): e = y - u, q = k^T (H0 k); T <- a1*beta*(P*T + w e); partial sums per sample
template <typename T>
__global__ __launch_bounds__(256) void svgp_mid_kernel(int64_t SB, int64_t B, int64_t M, int P, const T* __restrict__ Kuf,
                                                       T* __restrict__ Text, const T* __restrict__ Y, int64_t sY,
                                                       const T* __restrict__ w /* M x P */, const T* __restrict__ noise, double a1,
                                                       int want_grad, T* __restrict__ E /* SB x P */, T* __restrict__ dY, int dY_shared,
                                                       double* __restrict__ scal /* [S][2] : sum q, sum e^2 */) {
    __shared__ double red[16];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = n < SB;
    const int64_t s = valid ? n / B : 0, nb = valid ? n % B : 0;
    const T beta = (T)1 / noise[0];
    constexpr int PMAX = 8;
    T e[PMAX];
    double e2 = 0;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        e[p] = 0;
        if (p < P && valid) {
            e[p] = Y[s * sY + nb * P + p] - Text[(M + p) * SB + n];
            e2 += (double)e[p] * (double)e[p];
            if (E) E[n * P + p] = e[p];
            if (want_grad && dY) {
                const T g = (T)(-a1) * beta * e[p];
                if (dY_shared) atomic_add(dY + nb * P + p, g); else dY[n * P + p] = g;
            }
        }
    }
    T q = 0;
    const T c1 = (T)a1 * beta, cP = (T)P;
    if (valid) {
        for (int64_t m = 0; m < M; ++m) {
            const T k = Kuf[m * SB + n], t = Text[m * SB + n];
            q = fma(k, t, q);
            if (want_grad) {
                T we = 0;
#pragma unroll
                for (int p = 0; p < PMAX; ++p) if (p < P) we = fma(w[m * P + p], e[p], we);
                Text[m * SB + n] = c1 * (cP * t + we);
            }
        }
    }
    // blocks never straddle... they may: reduce per sample with a segmented approach (block spans <= 2 samples when B>=256;
    // general case: per-thread atomics when the block straddles)
    const int64_t n0 = (int64_t)blockIdx.x * 256, n1 = (n0 + 255 < SB - 1) ? n0 + 255 : SB - 1;
    if (n0 / B == n1 / B) {
        double qs = block_sum<double>((double)q, red);
        double es = block_sum<double>(e2, red);
        if (threadIdx.x == 0) { atomic_add(scal + 2 * (n0 / B), qs); atomic_add(scal + 2 * (n0 / B) + 1, es); }
    } else if (valid) {
        atomic_add(scal + 2 * s, (double)q);
        atomic_add(scal + 2 * s + 1, e2);
    }
}

// U[p][n] = sum_m w[m][p] * Kuf[m][n]  (the w^T row block of [H0; w^T] Kuf, kept out of the MFMA GEMM: a 1-row tile would waste
// a whole 128-row tile).  HBM-read bound (M*SB*e bytes), 16-byte loads, lanes <-> columns.
// gridDim.y > 1 (few columns: r05): the M rows are dealt to gridDim.y workgroups per column block, which ADD into U (zeroed by the caller) --
// with 1 024 columns the one-workgroup-per-512-columns form is two workgroups walking 1 024 dependent rows each: 0.37 ms of a 2 ms step
template <typename T, int PT>
__global__ __launch_bounds__(256) void wt_kuf_kernel(int64_t M, int64_t SB, int P, const T* __restrict__ Kuf, const T* __restrict__ w,
                                                     T* __restrict__ U) {
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    const int64_t n0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (n0 >= SB) return;
    if (gridDim.y > 1) {
        const int64_t mc = (M + gridDim.y - 1) / gridDim.y, mb = (int64_t)blockIdx.y * mc, me = mb + mc < M ? mb + mc : M;
        T a2[PT][VEC];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int v = 0; v < VEC; ++v) a2[p][v] = 0;
        for (int64_t m = mb; m < me; ++m)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T k = (n0 + v < SB) ? Kuf[m * SB + n0 + v] : (T)0;
#pragma unroll
                for (int p = 0; p < PT; ++p) a2[p][v] = fma((p < P) ? w[m * P + p] : (T)0, k, a2[p][v]);
            }
#pragma unroll
        for (int p = 0; p < PT; ++p)
            if (p < P)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (n0 + v < SB) atomic_add(U + (int64_t)p * SB + n0 + v, a2[p][v]);
        return;
    }
    T acc[PT][VEC];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[p][v] = 0;
    if (n0 + VEC <= SB && (SB % VEC) == 0) {
        for (int64_t m = 0; m < M; ++m) {
            const V kv = *reinterpret_cast<const V*>(Kuf + m * SB + n0);
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                const T wp = (p < P) ? w[m * P + p] : (T)0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[p][v] = fma(wp, kv[v], acc[p][v]);
            }
        }
    } else {
        for (int64_t m = 0; m < M; ++m)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T k = (n0 + v < SB) ? Kuf[m * SB + n0 + v] : (T)0;
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p][v] = fma((p < P) ? w[m * P + p] : (T)0, k, acc[p][v]);
            }
    }
#pragma unroll
    for (int p = 0; p < PT; ++p)
        if (p < P)
#pragma unroll
            for (int v = 0; v < VEC; ++v) if (n0 + v < SB) U[(int64_t)p * SB + n0 + v] = acc[p][v];
}

// The same row block from the split planes of Kfu (operand (n, k = m), gemm_split.hip layout): U[p][n] = sum_m w[m][p] Kfu[n][m].
// lane <-> column n (32 contiguous bytes per lane and plane per k block); HBM-read bound (4 bytes per element in the f16x2 format, 6 in bf16x3).
// NP = 2: f16x2 planes holding k / variance * 2^14 (result scaled by variance * 2^-14).
template <int PT, int NP>
__global__ __launch_bounds__(256) void wt_planes_kernel(int64_t M, int64_t SB, int P, const unsigned short* __restrict__ pl, int64_t pstride,
                                                        const float* __restrict__ w, float* __restrict__ U, const float* __restrict__ var) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= SB) return;
    float acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) acc[p] = 0.f;
    const int64_t K16 = (M + 15) / 16;
    for (int64_t kb = 0; kb < K16; ++kb) {
        const unsigned short* base = pl + (kb * SB + n) * 16;
        u4 v[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q) { v[q][0] = *reinterpret_cast<const u4*>(base + q * pstride); v[q][1] = *reinterpret_cast<const u4*>(base + q * pstride + 8); }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float x = 0.f;
#pragma unroll
            for (int q = NP - 1; q >= 0; --q) {
                const unsigned word = v[q][j >> 3][(j & 7) >> 1];
                if (NP == 2) {
                    x += (float)__builtin_bit_cast(_Float16, (unsigned short)((j & 1) ? (word >> 16) : (word & 0xffffu)));
                } else {
                    const unsigned bits = (j & 1) ? (word & 0xffff0000u) : (word << 16);
                    x += __builtin_bit_cast(float, bits);
                }
            }
            const int64_t m = kb * 16 + j;
            if (m < M) {
#pragma unroll
                for (int p = 0; p < PT; ++p) acc[p] = fma((p < P) ? w[m * P + p] : 0.f, x, acc[p]);
            }
        }
    }
    const float sc = NP == 2 ? var[0] * (1.f / 16384.f) : 1.f;
#pragma unroll
    for (int p = 0; p < PT; ++p) if (p < P) U[(int64_t)p * SB + n] = acc[p] * sc;
}

// A_Ki = (G - T1 - T1^T) + 1/2 (Gw mu^T + mu Gw^T) - b (P/2 Su + 1/2 mu mu^T)
__global__ void aki_kernel(int64_t M, int P, const double* __restrict__ G, const double* __restrict__ T1, const double* __restrict__ Gw,
                           const double* __restrict__ mu, const double* __restrict__ Su, double b, double* __restrict__ A) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * M; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / M, j = idx % M;
        double v = G[idx] - T1[idx] - T1[j * M + i];
        double r1 = 0, mm = 0;
        for (int p = 0; p < P; ++p) { if (Gw) r1 += Gw[i * P + p] * mu[j * P + p] + mu[i * P + p] * Gw[j * P + p]; mm += mu[i * P + p] * mu[j * P + p]; }
        A[idx] = v + 0.5 * r1 - b * (0.5 * P * Su[idx] + 0.5 * mm);
    }
}
// (r06) dSu = -X + c (Su^-1 - Ki)
__global__ void dsu_kernel(int64_t n, const double* __restrict__ X, const double* __restrict__ Sui, const double* __restrict__ Ki, double c, double* __restrict__ dSu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dSu[i] = c * (Sui[i] - Ki[i]) - X[i];
}
// (r06) dKuu0 = -X + Y + Y^T + b (P/2 KSK + 1/2 w w^T) - bP/2 Ki,  KSK = Ki Su Ki given or = Ki - H0
__global__ void dkuu0_kernel(int64_t M, int P, const double* __restrict__ X, const double* __restrict__ Y, const double* __restrict__ Ki, const double* __restrict__ KSK,
                             const double* __restrict__ H0, const double* __restrict__ w, double b, double* __restrict__ dKuu) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * M; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / M, j = idx % M;
        double ww = 0;
        for (int p = 0; p < P; ++p) ww += w[i * P + p] * w[j * P + p];
        const double ksk = KSK ? KSK[idx] : Ki[idx] - H0[idx];
        dKuu[idx] = -X[idx] + Y[idx] + Y[j * M + i] + b * (0.5 * P * ksk + 0.5 * ww) - 0.5 * b * P * Ki[idx];
    }
}
// (r06) Gw == nullptr above: the part A0 of A_Ki that does not depend on the reverse pass's R.  -Ki A_Ki Ki is linear in A_Ki and
// Ki (Gw mu^T) Ki = (Ki Gw)(Ki mu)^T = v w^T, so the two M^3 products run on A0 under the T product / the reverse pass and what is left for the
// step's tail is the rank-2P term:  dKuu -= 1/2 (v w^T + w v^T),  v = Ki Gw = dmu + b w  (dmu = Ki Gw - b w is formed anyway)
__global__ void dkuu_rank_kernel(int64_t M, int P, const double* __restrict__ dmu, const double* __restrict__ w, double b, double* __restrict__ dKuu) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * M; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / M, j = idx % M;
        double r = 0;
        for (int p = 0; p < P; ++p) {
            const double vi = dmu[i * P + p] + b * w[i * P + p], vj = dmu[j * P + p] + b * w[j * P + p];
            r += vi * w[j * P + p] + w[i * P + p] * vj;
        }
        dKuu[idx] -= 0.5 * r;
    }
}
// G' = (P/2 * a1 * beta) * Psi2 ; Gw = a1*beta*R   (beta on device)
template <typename T>
__global__ void scale_beta_kernel(int64_t n, const T* __restrict__ src, const double* __restrict__ noise, double c, double* __restrict__ dst) {
    const double beta = 1.0 / noise[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = c * beta * (double)src[i];
}
// logL[s], and the direct (non-Gram) gradient terms w.r.t. noise and kernel variance
template <typename T>
__global__ void svgp_finalize_kernel(int S, int64_t B, int64_t M, int P, const double* __restrict__ scal, const double* __restrict__ noise,
                                     const double* __restrict__ var, const double* __restrict__ sldL, const double* __restrict__ sldLs,
                                     const double* __restrict__ trKiSu, const double* __restrict__ muw, double scaling, double a1,
                                     T* __restrict__ logL, double* __restrict__ dnoise, double* __restrict__ dvar_direct,
                                     const double* __restrict__ qfix = nullptr) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double s2 = noise[0], beta = 1.0 / s2, vk = var[0];
    const double negKL = 0.5 * P * ((double)M + 2.0 * sldLs[0] - 2.0 * sldL[0] - trKiSu[0]) - 0.5 * muw[0];
    // whitened tier: sum_n q_n over ALL samples = tr((I - A_s A_s^T) Phi), formed in float64 from Phi = V V^T (qfix = [c tr Phi, c tr(Su L^-T Phi L^-1)],
    // c = P a1 beta / 2).  The reverse pass's own per-sample sums q_n = k_n . T_n multiply a recomputed float32 k_n by |T| ~ sqrt(cond): they
    // keep the split between the samples, their total is replaced (the mean over samples -- the only reduction the reference applies,
    // factor_graph.py:233 -- then carries the accurate total).
    double qshift = 0;
    if (qfix) {
        double qs = 0;
        for (int s = 0; s < S; ++s) qs += scal[2 * s];
        qshift = ((qfix[0] - qfix[1]) / (0.5 * P * a1 * beta) - qs) / (double)S;
    }
    double dn = 0;
    for (int s = 0; s < S; ++s) {
        const double Q0 = scal[2 * s] + qshift, E2 = scal[2 * s + 1];
        const double l = -0.5 * (double)B * P * (LOG2PI + log(s2)) - 0.5 * P * beta * (double)B * vk - 0.5 * beta * E2 + 0.5 * P * beta * Q0;
        logL[s] = (T)(scaling * l + negKL);
        dn += 0.5 * (double)B * P / beta - 0.5 * P * (double)B * vk - 0.5 * E2 + 0.5 * P * Q0;
    }
    if (dnoise) dnoise[0] = a1 * (-beta * beta) * dn;
    // (whitened tier: the reverse pass's kernel-variance gradient contains P beta sum_n q_n / variance through Kuf -- the same total, corrected the same way)
    if (dvar_direct) dvar_direct[0] = a1 * (double)S * (-0.5 * P * beta * (double)B) + a1 * P * beta * qshift * (double)S / vk;
}

// materialised-Gram mode (any kernel with an autograd-capable K on the host: Add / Multiply / Linear ... ): the caller passes
// Kuu (M x M, without jitter), Kuf (M x B), Kdiag (B) and receives d/dKuu, d/dKuf, d/dKdiag instead of kernel-parameter gradients
template <typename T>
struct SvgpMat { const T* Kuu = nullptr; const T* Kuf = nullptr; const T* Kdiag = nullptr; T* dKuu = nullptr; T* dKuf = nullptr; T* dKdiag = nullptr; };
__global__ void add_diag_kernel(int64_t n, double* __restrict__ A, double v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) A[i * n + i] += v;
}
// ---- heteroscedastic / per-output noise (svgp_regression.py:61-67): noise is (nrows, ncols), nrows in {1,B}, ncols in {1,P} ------
// one thread per column n: beta_d = 1/noise[n|0][d|0], bs = sum_d beta_d, e = y - u, q = k^T H0 k;
//   l_s += -1/2 sum_d (beta_d e_d^2 + log 2pi + log noise_d) - 1/2 bs (var - q)
// reverse: dY = -a1 beta.e ; Eb = a1 beta.e (for Gw = Kuf Eb) ; T[:,n] <- a1 (bs T[:,n] + w (beta.e)) = dKuf ; Ksc = 1/2 a1 bs Kuf[:,n]
// per-column quantities of one Y sample (shared by the two passes below).  ys = index of the Y sample: the column's own sample, or -- when
// X (hence Kuf) is shared and only Y is sampled -- one of the SY samples that share the column
template <typename T>
__device__ __forceinline__ void het_column(int64_t n, int64_t ys, int64_t SB, int64_t B, int64_t M, int P, const T* __restrict__ Text,
                                           const T* __restrict__ Y, int64_t sY, const T* __restrict__ noise, int64_t nrows, int ncols,
                                           double (&be)[8], double (&beta)[8], double& e2b, double& lg, double& bs) {
    const int64_t nb = n % B, nr = (nrows > 1) ? nb : 0;
    e2b = 0; lg = 0; bs = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        be[p] = 0; beta[p] = 0;
        if (p < P) {
            const double nz = (double)noise[nr * ncols + (ncols > 1 ? p : 0)];
            beta[p] = 1.0 / nz;
            const double e = (double)Y[ys * sY + nb * P + p] - (double)Text[(M + p) * SB + n];
            be[p] = beta[p] * e;
            e2b += be[p] * e; lg += LOG2PI + log(nz); bs += beta[p];
        }
    }
}
// pass 1, grid (column tiles, row chunks of RCH rows): q_n partial sums (atomics into qbuf) and, for the reverse mode, the in-place
// T[:,n] <- a1 (SY bs T[:,n] + w sum_ys(beta.e)) and Ksc = 1/2 a1 SY bs Kuf[:,n].  Rows are split over blockIdx.y so that M x S*B pairs fill
// the chip (one thread per column alone left it latency bound: 4 ms at 512 x 131072).
constexpr int HET_RCH = 32;
template <typename T>
__global__ __launch_bounds__(256) void svgp_het_rows_kernel(int64_t SB, int64_t B, int64_t M, int P, int SY, const T* __restrict__ Kuf,
                                                            T* __restrict__ Text, const T* __restrict__ Y, int64_t sY, const T* __restrict__ w,
                                                            const T* __restrict__ noise, int64_t nrows, int ncols, double a1, int want_grad,
                                                            T* __restrict__ Ksc, T* __restrict__ qbuf) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= SB) return;
    double be[8], beta[8], e2b, lg, bs, besum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ys = 0; ys < SY; ++ys) {
        het_column<T>(n, SY > 1 ? ys : n / B, SB, B, M, P, Text, Y, sY, noise, nrows, ncols, be, beta, e2b, lg, bs);
#pragma unroll
        for (int p = 0; p < 8; ++p) besum[p] += be[p];
    }
    const double bst = bs * (double)SY;
    const int64_t m0 = (int64_t)blockIdx.y * HET_RCH, m1 = (m0 + HET_RCH < M) ? m0 + HET_RCH : M;
    double q = 0;
    for (int64_t m = m0; m < m1; ++m) {
        const double k = (double)Kuf[m * SB + n], t = (double)Text[m * SB + n];
        q = fma(k, t, q);
        if (want_grad) {
            double we = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) if (p < P) we = fma((double)w[m * P + p], besum[p], we);
            Text[m * SB + n] = (T)(a1 * (bst * t + we));
            Ksc[m * SB + n] = (T)(0.5 * a1 * bst * k);
        }
    }
    atomic_add(qbuf + n, (T)q);
}
// pass 2, one thread per column: l_s, sum bs, dY, Eb, dnoise, dKdiag.  Sums that land on ONE address (the per-sample scalars; dnoise when the
// noise is shared by all rows) are reduced over the block first -- 131 072 same-address float64 atomics cost 4 ms.
template <typename T>
__global__ __launch_bounds__(256) void svgp_het_cols_kernel(int64_t SB, int64_t B, int64_t M, int P, int SY, const T* __restrict__ Text,
                                                            const T* __restrict__ Y, int64_t sY, const T* __restrict__ noise, int64_t nrows,
                                                            int ncols, const double* __restrict__ var,
                                                            const T* __restrict__ kdiag /* per column, or NULL -> var[0] */, double a1,
                                                            int want_grad, const T* __restrict__ qbuf, T* __restrict__ Eb, T* __restrict__ dY,
                                                            int dY_shared, T* __restrict__ dnoise, T* __restrict__ dkdiag,
                                                            double* __restrict__ scal /* [S][2]: l_s, sum bs */) {
    __shared__ double red[16];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = n < SB;
    const int64_t nn = valid ? n : SB - 1;
    const int64_t s = nn / B, nb = nn % B, nr = (nrows > 1) ? nb : 0;
    const double q = (double)qbuf[nn];
    const double vk = kdiag ? (double)kdiag[nn] : var[0];
    const int64_t n0 = (int64_t)blockIdx.x * 256, n1 = (n0 + 255 < SB - 1) ? n0 + 255 : SB - 1;
    const bool one_sample = (n0 / B) == (n1 / B);
    double besum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bs = 0;
    for (int ys = 0; ys < SY; ++ys) {
        double be[8], beta[8], e2b, lg;
        // NOTE: U (rows M.. of Text) is untouched by pass 1, so e is recomputed from the same values
        const int64_t sidx = SY > 1 ? ys : s;
        het_column<T>(nn, sidx, SB, B, M, P, Text, Y, sY, noise, nrows, ncols, be, beta, e2b, lg, bs);
        const double lval = valid ? -0.5 * (e2b + lg) - 0.5 * bs * (vk - q) : 0.0, bval = valid ? bs : 0.0;
        if (one_sample || SY > 1) {
            const double ls_ = block_sum<double>(lval, red), bb_ = block_sum<double>(bval, red);
            const int64_t sw = SY > 1 ? ys : n0 / B;
            if (threadIdx.x == 0) { atomic_add(scal + 2 * sw, ls_); atomic_add(scal + 2 * sw + 1, bb_); }
        } else if (valid) {
            atomic_add(scal + 2 * s, lval);
            atomic_add(scal + 2 * s + 1, bval);
        }
        if (want_grad) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (p < P) {
                    besum[p] += be[p];
                    gn[p] += a1 * (0.5 * be[p] * be[p] - 0.5 * beta[p] + 0.5 * (vk - q) * beta[p] * beta[p]);
                    if (valid && dY) {
                        const T g = (T)(-a1 * be[p]);
                        if (SY > 1) dY[(ys * B + nb) * P + p] = g;
                        else if (dY_shared) atomic_add(dY + nb * P + p, g);
                        else dY[n * P + p] = g;
                    }
                }
        }
    }
    if (!want_grad) return;
    if (valid && dkdiag) dkdiag[n] = (T)(-0.5 * a1 * bs * (double)SY);
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if (p < P) {
            if (valid) Eb[n * P + p] = (T)(a1 * besum[p]);
            if (dnoise) {
                const double g = valid ? gn[p] : 0.0;
                if (nrows > 1) { if (valid) atomic_add(dnoise + nr * ncols + (ncols > 1 ? p : 0), (T)g); }
                else {
                    const double gs = block_sum<double>(g, red);          // shared noise: one address per output column
                    if (threadIdx.x == 0) atomic_add(dnoise + (ncols > 1 ? p : 0), (T)gs);
                }
            }
        }
}
template <typename T>
__global__ void svgp_het_finalize_kernel(int S, int64_t M, int P, const double* __restrict__ scal, const double* __restrict__ sldL,
                                         const double* __restrict__ sldLs, const double* __restrict__ trKiSu, const double* __restrict__ muw,
                                         double scaling, double a1, T* __restrict__ logL, double* __restrict__ dvar_direct) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double negKL = 0.5 * P * ((double)M + 2.0 * sldLs[0] - 2.0 * sldL[0] - trKiSu[0]) - 0.5 * muw[0];
    double sb = 0;
    for (int s = 0; s < S; ++s) { logL[s] = (T)(scaling * scal[2 * s] + negKL); sb += scal[2 * s + 1]; }
    if (dvar_direct) dvar_direct[0] = -0.5 * a1 * sb;
}

// ---- streaming heteroscedastic bound (r04): per-row noise (B, 1), one output column, float32 split path ------------------------------------
// svgp_regression.py:61-67 with noise (N, 1): beta_n = 1 / noise_n.  With nmin = min_n noise_n and the weights r_n = nmin beta_n <= 1:
//   planes of Kuf diag(sqrt r)  ->  Psi2' = nmin sum_n beta_n k_n k_n^T           (the core's G = P a1 / 2 (1/nmin) Psi2': the homoscedastic code with noise := nmin)
//   planes of diag(r) Kfu (+ U' = r U)  ->  T'' = T diag(r),  y' = r y            (the fused reverse pass with noise := nmin sees e'' = r e and forms exactly
//                                                                                  a1 beta_n (w e_n + P T_n), sum_n beta_n q_n, dY = -a1 beta_n e_n, R = Kuf (beta e))
// What the fused pass cannot deliver -- sum_n beta_n e_n^2 and the per-row noise gradient, which needs q_n per COLUMN -- comes from one
// extra pass over the Kfu planes and T'' (het_stats_kernel).
__global__ __launch_bounds__(256) void het_min_kernel(int64_t B, const float* __restrict__ noise, unsigned* __restrict__ minbits) {
    __shared__ unsigned wm[4];
    unsigned m = 0x7f800000u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) { const unsigned b = __builtin_bit_cast(unsigned, noise[i]); m = b < m ? b : m; }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t < m ? t : m; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < 4; ++w) m = wm[w] < m ? wm[w] : m; atomicMin(minbits, m); }      // positive floats order as their bits
}
__global__ __launch_bounds__(256) void het_prep_kernel(int64_t B, const float* __restrict__ noise, const unsigned* __restrict__ minbits, float* __restrict__ cs,
                                                       float* __restrict__ rs, double* __restrict__ hs /* [0] sum log noise, [1] sum 1/noise */,
                                                       float* __restrict__ nzf, double* __restrict__ noised) {
    __shared__ double red[16];
    const float nmin = __builtin_bit_cast(float, minbits[0]);
    double sl = 0, sb = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        const float nz = noise[i], r = nmin / nz;
        rs[i] = r; cs[i] = sqrtf(r);
        sl += log((double)nz); sb += 1.0 / (double)nz;
    }
    sl = block_sum<double>(sl, red); sb = block_sum<double>(sb, red);
    if (threadIdx.x == 0) { atomic_add(hs + 0, sl); atomic_add(hs + 1, sb); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { nzf[0] = nmin; noised[0] = (double)nmin; }
}
__global__ void het_scale_y_kernel(int64_t n, int64_t B, const float* __restrict__ Y, const float* __restrict__ rs, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = Y[i] * rs[i % B];
}
// one wave per 16-column block: lane = (column c = lane % 16, row phase lane / 16); rows m = phase, phase + 4, ...: the T'' reads of one step
// are 4 rows x 64 bytes contiguous.  q_n = (sum_m kpl_mn T''_mn) sigma^2 2^-14 / r_n^2, e_n = (y'_n - U'_n) / r_n.
__global__ __launch_bounds__(256) void het_stats_kernel(int64_t SB, int64_t B, int64_t M, const unsigned short* __restrict__ pl, int64_t pstride,
                                                        const float* __restrict__ Tb /* T'' in 16-column blocks, then U' at Tb + M * SB */,
                                                        const float* __restrict__ ysc, int64_t sY, const float* __restrict__ noise,
                                                        const float* __restrict__ rs, const float* __restrict__ var, double a1, int want_grad,
                                                        float* __restrict__ dnoise, double* __restrict__ sbe2 /* [S] */) {
    const int lane = threadIdx.x & 63, c = lane & 15, ph = lane >> 4;
    const int64_t nb16 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (nb16 * 16 >= SB) return;
    const int64_t n = nb16 * 16 + c;
    float acc = 0.f;
    typedef unsigned int hs_u32x2 __attribute__((ext_vector_type(2)));
    for (int64_t mb = 0; mb < M; mb += 16) {          // (M % 16 == 0) rows mb + 4 ph .. + 3: one 8-byte piece of each plane, four T'' rows
        const int64_t off = ((mb >> 4) * SB + n) * 16 + 4 * ph;
        const hs_u32x2 vh = *reinterpret_cast<const hs_u32x2*>(pl + off), vl = *reinterpret_cast<const hs_u32x2*>(pl + pstride + off);
        const float* tp = Tb + (nb16 * M + mb + 4 * ph) * 16 + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned wh = vh[i >> 1], wl = vl[i >> 1];
            const unsigned short hb = (unsigned short)((i & 1) ? (wh >> 16) : (wh & 0xffffu)), lb = (unsigned short)((i & 1) ? (wl >> 16) : (wl & 0xffffu));
            acc = fmaf((float)__builtin_bit_cast(_Float16, hb) + (float)__builtin_bit_cast(_Float16, lb), tp[i * 16], acc);
        }
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    const int64_t nbr = n % B, s = n / B;
    const double r = (double)rs[nbr], nz = (double)noise[nbr], beta = 1.0 / nz, vk = (double)var[0];
    const double q = (double)acc * vk * (1.0 / 16384.0) / (r * r);
    const double e = ((double)ysc[s * sY + nbr] - (double)Tb[M * SB + n]) / r;
    double be2 = (ph == 0) ? beta * e * e : 0.0;
    // the 16 columns of a block lie in one sample when B % 16 == 0 (host guarantees it): one atomic per wave
    for (int o = 8; o > 0; o >>= 1) be2 += __shfl_xor(be2, o, 64);
    if (lane == 0) atomic_add(sbe2 + s, be2);
    if (want_grad && dnoise && ph == 0) atomic_add(dnoise + nbr, (float)(a1 * (0.5 * beta * beta * e * e - 0.5 * beta + 0.5 * (vk - q) * beta * beta)));
}
template <typename T>
__global__ void svgp_finalize_hets_kernel(int S, int64_t B, int64_t M, const double* __restrict__ scal, const double* __restrict__ sbe2,
                                          const double* __restrict__ hs, const double* __restrict__ noised /* nmin */, const double* __restrict__ var,
                                          const double* __restrict__ sldL, const double* __restrict__ sldLs, const double* __restrict__ trKiSu,
                                          const double* __restrict__ muw, double scaling, double a1, T* __restrict__ logL, double* __restrict__ dvar_direct) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double vk = var[0], bmax = 1.0 / noised[0];
    const double negKL = 0.5 * ((double)M + 2.0 * sldLs[0] - 2.0 * sldL[0] - trKiSu[0]) - 0.5 * muw[0];
    for (int s = 0; s < S; ++s) {
        const double l = -0.5 * (sbe2[s] + (double)B * LOG2PI + hs[0]) - 0.5 * vk * hs[1] + 0.5 * bmax * scal[2 * s];
        logL[s] = (T)(scaling * l + negKL);
    }
    if (dvar_direct) dvar_direct[0] = a1 * (double)S * (-0.5 * hs[1]);
}

// |A|_1 of a symmetric (n x n) float64 matrix: max over rows of the absolute row sum (= column sum), into *out as a double (non-negative
// doubles order like unsigned 64-bit integers: one atomicMax per workgroup; *out must be zeroed)
__global__ __launch_bounds__(256) void norm1_sym_kernel(int64_t n, const double* __restrict__ A, int64_t lda, double* __restrict__ out) {
    __shared__ double red[16];
    const int64_t row = blockIdx.x;
    double s = 0;
    for (int64_t j = threadIdx.x; j < n; j += 256) s += fabs(A[row * lda + j]);
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), __builtin_bit_cast(unsigned long long, s));
}

// the training call's first launch on the caller's stream: its status words (LAPACK info of the two factorisations, the split scale word)
// cleared in ONE launch instead of a hipMemsetAsync each in front of the kernels of the critical chain (~10 us apiece there)
__global__ void svgp_init_kernel(int* __restrict__ info, int* __restrict__ info2, double* __restrict__ cond_dev) {
    if (threadIdx.x == 0 && info) info[0] = 0;
    if (threadIdx.x < 8) info2[threadIdx.x] = 0;
    // the condition accumulators of THIS call (its norm kernels fold in with atomicMax): cleared here and not only by the previous call's
    // last launch -- a call that returned early (an error between its norm kernels and cond_publish_kernel) must not leak its norms
    if (threadIdx.x < 2) cond_dev[threadIdx.x] = 0.0;
}
// the same launch with the float64 copies of Z, the length-scales and the variance the core works on (and sigma for the whitened tier): three
// more dependent launches in front of the Kuu Gram otherwise
template <typename T>
__global__ __launch_bounds__(256) void svgp_prologue_kernel(int* __restrict__ info, int* __restrict__ info2, double* __restrict__ cond_dev, int64_t nZ,
                                                            const T* __restrict__ Z, double* __restrict__ Zd, int64_t nls, const T* __restrict__ ls,
                                                            double* __restrict__ lsd, const T* __restrict__ var, double* __restrict__ vard,
                                                            float* __restrict__ sig) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0 && info) info[0] = 0;
        if (threadIdx.x < 8) info2[threadIdx.x] = 0;
        if (threadIdx.x < 2) cond_dev[threadIdx.x] = 0.0;
        if (threadIdx.x == 0) { vard[0] = (double)var[0]; if (sig) sig[0] = sqrtf((float)var[0]); }
        for (int64_t i = threadIdx.x; i < nls; i += blockDim.x) lsd[i] = (double)ls[i];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nZ; i += (int64_t)gridDim.x * blockDim.x) Zd[i] = (double)Z[i];
}
// sigma = sqrt(variance) as a float word (the whitened tier's planes hold V / sigma * 2^14: |v_n|^2 <= k_nn = variance)
__global__ void svgp_sigma_kernel(const float* __restrict__ var, float* __restrict__ sig) { if (threadIdx.x == 0) sig[0] = sqrtf(var[0]); }

// the training call's last launch: cond_1(Kuu + jitter I) = |K|_1 |K^-1|_1 of this call folded into the running maximum the host can read
// without synchronising (pinned, device-visible memory; mxf_svgp_cond_nowait)
__global__ void cond_publish_kernel(double* __restrict__ cond_dev, double* __restrict__ host_slot /* [0] running max, [1] last */) {
    const double c = cond_dev[0] * cond_dev[1];
    if (c > host_slot[0]) host_slot[0] = c;
    host_slot[1] = c;
    __threadfence_system();
    cond_dev[2] = cond_dev[0]; cond_dev[3] = cond_dev[1];       // kept for mxf_svgp_last_cond
    cond_dev[0] = 0.0; cond_dev[1] = 0.0;                       // the next call's norm kernels accumulate with atomicMax: no memset in front of them
}

// |A|_1 (see norm1_sym_kernel) with 16 rows per workgroup -- one row per wave at a time (coalesced loads + a DPP wave sum, no barrier per
// row) and ONE atomic per workgroup: n / 16 same-address atomics instead of n (they serialise: 1024 of them were most of the 80 us this
// took on the critical path in front of the Cholesky factorisation).
__global__ __launch_bounds__(256) void norm1_sym16_kernel(int64_t n, const double* __restrict__ A, int64_t lda, double* __restrict__ out) {
    __shared__ double wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double best = 0;
    for (int r = wave; r < 16; r += 4) {
        const int64_t row = (int64_t)blockIdx.x * 16 + r;
        if (row >= n) break;
        double s = 0;
        for (int64_t j = lane; j < n; j += 64) s += fabs(A[row * lda + j]);
        s = wave_sum(s);
        best = s > best ? s : best;
    }
    if (lane == 0) wmax[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = wmax[w] > best ? wmax[w] : best;
        atomicMax(reinterpret_cast<unsigned long long*>(out), __builtin_bit_cast(unsigned long long, best));
    }
}

template <typename T>
int svgp_logpdf_typed(mxf_ctx* h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P, const T* X, int64_t sX, const T* Y,
                      int64_t sY, const T* Z, const T* noise, int64_t nrows, int ncols, const T* mu, const T* W, const T* sdiag, const T* ls, int ard, const T* var,
                      double jitter, double scaling, double gscale, T* logL, int* info, int want_grad, T* dX, T* dY, T* dZ, T* dnoise,
                      T* dmu, T* dW, T* dSdiag, T* dls, T* dvar, hipStream_t st, SvgpMat<T> mat = SvgpMat<T>()) {
    if (P > 8) MXF_FAIL(h, -3, "mxf_svgp_logpdf: P > 8 outputs not supported");
    if ((nrows != 1 && nrows != B) || (ncols != 1 && ncols != P)) MXF_FAIL(h, -2, "mxf_svgp_logpdf: noise_var must be (1|B, 1|P)");
    const bool use_mat = mat.Kuu != nullptr;
    // sampled Y over shared X (hence shared Kuf, T, U): the S samples share the B columns -- generic path, the data term is quadratic in Y
    const bool ysamp = S > 1 && sX == 0 && sY != 0;
    if (ysamp && sY != B * P) MXF_FAIL(h, -2, "mxf_svgp_logpdf: Y samples must be contiguous");
    // generic path: Kuf-side reverse mode through a materialised dKuf (not the streaming fused pass); also for Q > 16 inputs, which the
    // register-tiled fused reverse pass does not cover (gram_bwd.hip: generic kernel)
    // r04: per-row noise (B, 1) with one output column on the float32 split path runs the STREAMING form (het_* kernels above); every other
    // heteroscedastic shape keeps the generic (materialised dKuf) path
    static const int het_stream_env = MXF_KNOB("MXF_SVGP_HET_STREAM", 1);
    // float32 streaming: the two big GEMMs run on the 16-bit matrix pipe from split planes of their operands (gemm_split.hip)
    static const int split_env = MXF_KNOB("MXF_SVGP_SPLIT", 1);
    const int64_t SBh = ((sX == 0) ? (int64_t)1 : (int64_t)S) * B;
    const bool het_stream = (het_stream_env != 0) && (split_env != 0) && sizeof(T) == 4 && want_grad && nrows == B && nrows > 1 && ncols == 1 && P == 1 && !use_mat && !ysamp && Q <= 8 &&
                            (B % 16 == 0) && (M % 16 == 0) && M >= 128 && h->svgp_form == MXF_SVGP_EXPLICIT && mxf_svgp_bwd_is_mfma(kind, dtype, SBh, B, Q, P, X);
    const bool het = (nrows > 1 || ncols > 1 || use_mat || ysamp || Q > 16) && !het_stream;
    if (sX != 0 && sX != B * Q) MXF_FAIL(h, -2, "mxf_svgp_logpdf: X samples must be contiguous");
    if (S > 1 && sX == 0 && sY == 0) MXF_FAIL(h, -3, "mxf_svgp_logpdf: S > 1 with neither X nor Y sampled");
    const int SS = (sX == 0) ? 1 : S;   // samples that need their own columns
    const int SYc = ysamp ? S : 1;     // Y samples per column
    const int64_t SB = (int64_t)SS * B, MM = M * M, MP = M * P;
    const int lsn = ard ? Q : 1;
    const double a1 = gscale * scaling, bw = gscale * (double)S;
    typedef double D;

    size_t need = 0;
    auto acc = [&](size_t n, size_t es) { need += mxf_align(n * es); };
    acc(M * Q, 8); acc(lsn, 8); acc(1, 8); acc(1, 8); acc(MP, 8); acc(MM, 8); acc(M, 8);   // f64 copies of the parameters
    for (int i = 0; i < 9; ++i) acc(MM, 8);   // L, Linv, Ki, Su(Ls), Lsinv, Sui, KiSu, H0, tmp
    acc(MP, 8); acc(16, 8); acc(2 * (size_t)S, 8); acc(8, sizeof(int));
    acc((size_t)(M + P) * M, sizeof(T)); acc(MP, sizeof(T));
    acc((size_t)(M + P) * SB, sizeof(T)); acc((size_t)SB, sizeof(T));
    const bool use_split = split_env && want_grad && sizeof(T) == 4 && !het && (SB % 16 == 0) && (M % 16 == 0) && M >= 128 && Q <= 16;
    // operand format of the split GEMMs: two scaled f16 terms / three products (default) or three bf16 terms / six products
    static const int split_mode = MXF_KNOB("MXF_SPLIT_BF16X3", 0) ? MXF_SPLIT_BF16X3 : MXF_SPLIT_F16X2;
    const float split_ga = split_mode == MXF_SPLIT_F16X2 ? (1.f / 16384.f) : 1.f;     // Gram planes hold k / variance * 2^14 in the f16x2 format
    const float* split_var = split_mode == MXF_SPLIT_F16X2 ? (const float*)var : nullptr;
    const size_t pl_big = mxf_split_plane_elems(M, SB), pl_h0 = mxf_split_plane_elems(M, M);     // == mxf_split_plane_elems(SB, M)
    const size_t gp_scr = use_split ? mxf_gram_planes_scratch_bytes(SB, SB, Q) : 0;      // upper bound for either orientation
    // float32 streaming form (mxf_svgp_configure): the whitened tier runs on the f16x2 split kernels' wide forms only
    const bool whiten = sizeof(T) == 4 && want_grad && h->svgp_form == MXF_SVGP_WHITENED;
    if (whiten && !(use_split && split_mode == MXF_SPLIT_F16X2 && (M % 128) == 0 && (SB % 256) == 0))
        MXF_FAIL(h, -3, "mxf_svgp_logpdf: the whitened float32 form needs M %% 128 == 0, S B %% 256 == 0, Q <= 16, homoscedastic noise (see mxf_svgp_whitened_ok)");
    // (each branch mirrors one carve below, in the same order)
    if (use_split) { acc(3 * pl_big, 2); acc(3 * pl_h0, 2); acc(3 * pl_big, 2); acc(gp_scr, 1); acc(gp_scr, 1); }
    else { acc((size_t)M * SB, sizeof(T)); if (want_grad) acc((size_t)M * SB, sizeof(T)); }      // Kuf, Kfu in the streaming dtype
    if (whiten) { acc(2 * pl_h0, 2); acc(MP, 8); acc(MP, sizeof(T)); acc(4, sizeof(float)); acc((size_t)(M / 128) * SB, sizeof(float)); }
    // r06: ONE set of Kuf planes.  T = H0 Kuf reads the planes Psi2 reads (operand (m, k = n)) through gemm_bt.hip's transposing LDS read, and
    // forms the row U = w^T Kuf on the way: the second planes pass (8.6 GB written between the two products, 1.8 ms of the 24 ms step with the
    // matrix pipe idle) and its 8.6 GB buffer are gone.  Needs the explicit float32 form, one output column, whole 256 x 256 tiles.
    static const int bt_env = (int)MXF_KNOB("MXF_SVGP_BT", 1);
    const bool bt_path = bt_env && use_split && !whiten && !het_stream && P == 1 && split_mode == MXF_SPLIT_F16X2 && M <= 2048 && mxf_gemm_bt_ok(M, SB, M);
    // the whitened tier likewise: T = Hh V reads the planes of V that Phi = V V^T reads (the V product's planes output IS the K-major layout),
    // and forms U = a^T V on the way -- the V product no longer writes the planes of V^T (8.6 GB, its "second output") nor the partial sums of U
    const bool bt_wh = bt_env && whiten && P == 1 && M <= 2048 && mxf_gemm_bt_ok(M, SB, M);
    if (bt_path || bt_wh) acc(2 * (size_t)M + 8, 2);
    if (het_stream) { acc(B, 4); acc(B, 4); acc((size_t)(sY == 0 ? B : SB), 4); acc(4, 8); acc(S, 8); acc(4, 4); }
    // r06: the generic (materialised-Gram / heteroscedastic) float32 path's two big products on the f16 matrix pipe as well: T = H0 Kuf through the
    // K-major product from the planes of Kuf (split from the float32 Gram: one maxabs + one split pass, 0.17 ms at 512 x 131 072), and
    // G' = Ksc Kuf^T from the planes of Ksc and the same Kuf planes -- f32-equivalent like the streaming path's products (three f16 products, f32
    // accumulation) where the generic kernel ran true-f32 MFMAs at 100 TF: the deep GP's first layer 0.68 + 0.66 ms -> planes 0.35 + products 0.3.
    static const int het_split_env = (int)MXF_KNOB("MXF_SVGP_HET_SPLIT", 1);
    const bool het_split = het_split_env && het && sizeof(T) == 4 && want_grad && split_env && split_mode == MXF_SPLIT_F16X2 && mxf_gemm_bt_ok(M, SB, M) &&
                           (int64_t)M * SB >= (int64_t)1 << 24;
    if (het_split) { acc(2 * pl_h0, 2); acc(2 * pl_big, 2); acc(2 * pl_big, 2); acc(4, sizeof(unsigned)); }
    if (want_grad) { acc(MM, sizeof(T)); acc(MP, sizeof(T)); acc((size_t)SB * P, sizeof(T)); for (int i = 0; i < 6; ++i) acc(MM, 8); acc(MP, 8); acc(MP, 8); acc(M * Q, 8); acc(lsn, 8); acc(4, 8); }
    void* ws = mxf_ws(h, need);
    if (!ws) MXF_FAIL(h, -4, "mxf_svgp_logpdf: cannot allocate %zu bytes of scratch", need);
    Carver cv(ws);
    D* Zd = cv.take<D>(M * Q); D* lsd = cv.take<D>(lsn); D* vard = cv.take<D>(1); D* noised = cv.take<D>(1);
    D* mud = cv.take<D>(MP); D* Wd = cv.take<D>(MM); D* sd = cv.take<D>(M);
    D* Lm = cv.take<D>(MM); D* Linv = cv.take<D>(MM); D* Ki = cv.take<D>(MM); D* Su = cv.take<D>(MM); D* Lsinv = cv.take<D>(MM);
    D* Sui = cv.take<D>(MM); D* KiSu = cv.take<D>(MM); D* H0 = cv.take<D>(MM); D* tmp = cv.take<D>(MM);
    D* wd = cv.take<D>(MP); D* sc = cv.take<D>(16); D* scal = cv.take<D>(2 * (size_t)S); int* info2 = cv.take<int>(8);
    T* Aext = cv.take<T>((size_t)(M + P) * M); T* wT = cv.take<T>(MP);
    T* Text = cv.take<T>((size_t)(M + P) * SB); T* qbuf = cv.take<T>((size_t)SB);
    unsigned short* plKfu = nullptr; unsigned short* plH0 = nullptr; unsigned short* plKuf = nullptr;
    T* Kuf = nullptr; T* Kfu = nullptr; T* Psi2 = nullptr; T* R = nullptr; T* Eb = nullptr;
    float* gscr0 = nullptr; float* gscr1 = nullptr;
    if (use_split) { plKfu = cv.take<unsigned short>(3 * pl_big); plH0 = cv.take<unsigned short>(3 * pl_h0); plKuf = cv.take<unsigned short>(3 * pl_big);
                     gscr0 = (float*)cv.take<char>(gp_scr); gscr1 = (float*)cv.take<char>(gp_scr); }
    else { Kuf = cv.take<T>((size_t)M * SB); if (want_grad) Kfu = cv.take<T>((size_t)M * SB); }
    unsigned short* plLi = nullptr; D* ad = nullptr; T* aT = nullptr; float* sigf = nullptr;
    float* upart = nullptr;
    if (whiten) { plLi = cv.take<unsigned short>(2 * pl_h0); ad = cv.take<D>(MP); aT = cv.take<T>(MP); sigf = cv.take<float>(4); upart = cv.take<float>((size_t)(M / 128) * SB); }
    unsigned short* wpl = nullptr;
    if (bt_path || bt_wh) wpl = cv.take<unsigned short>(2 * (size_t)M + 8);
    // whitened tier: the planes of V^T (operand (n, k = m) of T = Hh V) go into the THIRD plane slots of the two big buffers (sized for the
    // three-plane bf16 format; the whitened tier runs two-plane f16x2 only) -- the V product writes them next to V's own planes while other
    // workgroups still read the Kfu planes, so they cannot share that buffer's first two slots
    unsigned short* plVt = whiten ? plKfu + 2 * pl_big : nullptr;
    const int64_t pVt = whiten ? (int64_t)((plKuf + 2 * pl_big) - plVt) : 0;
    float* hcs = nullptr; float* hrs = nullptr; float* hys = nullptr; D* hhs = nullptr; D* hbe2 = nullptr; float* hnz = nullptr;
    if (het_stream) { hcs = cv.take<float>(B); hrs = cv.take<float>(B); hys = cv.take<float>((size_t)(sY == 0 ? B : SB)); hhs = cv.take<D>(4); hbe2 = cv.take<D>(S); hnz = cv.take<float>(4); }
    unsigned short* hsA = nullptr; unsigned short* hsK = nullptr; unsigned short* hsS = nullptr; unsigned* hsw = nullptr;
    if (het_split) { hsA = cv.take<unsigned short>(2 * pl_h0); hsK = cv.take<unsigned short>(2 * pl_big); hsS = cv.take<unsigned short>(2 * pl_big); hsw = cv.take<unsigned>(4); }
    if (want_grad) { Psi2 = cv.take<T>(MM); R = cv.take<T>(MP); Eb = cv.take<T>((size_t)SB * P); }
    D* G = nullptr; D* T1 = nullptr; D* AKi = nullptr; D* T2 = nullptr; D* dKuu = nullptr; D* dSu = nullptr;
    D* Gw = nullptr; D* dmud = nullptr; D* dZc = nullptr; D* dlsc = nullptr; D* dvc = nullptr;
    if (want_grad) {
        G = cv.take<D>(MM); T1 = cv.take<D>(MM); AKi = cv.take<D>(MM); T2 = cv.take<D>(MM); dKuu = cv.take<D>(MM); dSu = cv.take<D>(MM);
        Gw = cv.take<D>(MP); dmud = cv.take<D>(MP); dZc = cv.take<D>(M * Q); dlsc = cv.take<D>(lsn); dvc = cv.take<D>(4);
    }
    // sc: [0]=sumlogdiag L, [1]=sumlogdiag Ls, [2]=tr(Ki Su), [3]=mu.w, [4]=dnoise, [5]=dvar_direct
    // the accounting above and the carve must stay in step (a dangling else once had them 4 GB apart)
    if (cv.off > need) MXF_FAIL(h, -6, "mxf_svgp_logpdf: internal error, scratch carve %zu exceeds its accounting %zu", cv.off, need);
#ifdef MXF_PROBES
    if (cv.off != need) MXF_FAIL(h, -6, "mxf_svgp_logpdf: internal error, scratch carve %zu != accounting %zu", cv.off, need);
#endif

#define CONV(n, src, dst) hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(n)), dim3(256), 0, st, (int64_t)1, (int64_t)(n), src, (int64_t)(n), dst, (int64_t)(n))
    // (W and diag(s) are converted on the second side stream, where Su is formed: two launches less in front of the Kuu chain)
    // (everything the Kuu chain does not need itself -- noise, mu, W, diag(s), the scalar accumulators -- is prepared on the second side
    //  stream, where Su is formed; the main stream waits for that stream's ev_su before it first touches them)
    MXF_STAGE(h, "start", st);
    if (!mxf_cond_init(h)) MXF_FAIL(h, -4, "mxf_svgp_logpdf: cannot allocate the condition words");
    double* cond_slot = h->cond_host + 2 * h->cond_slot;
    for (int i = 0; i < MXF_NT; ++i) h->tm.used[i] = false;
    MXF_T0(h, MXF_T_CALL, st); MXF_T0(h, MXF_T_CHAIN, st);
    if (!use_mat)
        hipLaunchKernelGGL((svgp_prologue_kernel<T>), dim3(gridn(M * Q) > 64 ? 64 : gridn(M * Q)), dim3(256), 0, st, info, info2, h->cond_dev, (int64_t)(M * Q), Z, Zd,
                           (int64_t)lsn, ls, lsd, var, vard, whiten ? sigf : (float*)nullptr);
    else {
        hipLaunchKernelGGL(svgp_init_kernel, dim3(1), dim3(64), 0, st, info, info2, h->cond_dev);
        if (whiten) hipLaunchKernelGGL(svgp_sigma_kernel, dim3(1), dim3(64), 0, st, (const float*)var, sigf);
    }
    if (het_stream) {       // nmin, the row weights, sum log noise / sum beta, y' = r y -- before the fork: both side streams read them
        MXF_HIP(h, hipMemsetAsync(info2 + 5, 0x7f, sizeof(int), st));                 // 0x7f7f7f7f: a huge finite float, above any noise variance
        MXF_HIP(h, hipMemsetAsync(hhs, 0, 4 * sizeof(D), st));
        MXF_HIP(h, hipMemsetAsync(hbe2, 0, (size_t)S * sizeof(D), st));
        const unsigned pg = (unsigned)((B + 255) / 256 > 256 ? 256 : (B + 255) / 256);
        hipLaunchKernelGGL(het_min_kernel, dim3(pg), dim3(256), 0, st, B, (const float*)noise, (unsigned*)(info2 + 5));
        hipLaunchKernelGGL(het_prep_kernel, dim3(pg), dim3(256), 0, st, B, (const float*)noise, (const unsigned*)(info2 + 5), hcs, hrs, hhs, hnz, noised);
        const int64_t ny = sY == 0 ? B : SB;
        hipLaunchKernelGGL(het_scale_y_kernel, dim3(gridn(ny)), dim3(256), 0, st, ny, B, (const float*)Y, (const float*)hrs, hys);
    }
#undef CONV
    int rc;
    static const int64_t psi2_ka = MXF_KNOB("MXF_SVGP_PSI2_KA", -1);
    static const bool psi2_ra_env = MXF_KNOB_SET("MXF_SVGP_PSI2_RA");
    static const int psi2_ra = (int)MXF_KNOB("MXF_SVGP_PSI2_RA", 148);
    static const int psi2_rb = MXF_KNOB("MXF_SVGP_PSI2_RB", 16);
    // ---- core, float64, once; two independent chains run concurrently (main: Kuu -> L -> Ki, w; side: Kuf_all, Su -> Ls -> Su^-1) ----
    if (!mxf_side_init(h)) MXF_FAIL(h, -5, "mxf_svgp_logpdf: cannot create the internal side stream");
    hipStream_t sd_ = h->side;
    if (use_mat) {
        hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(MM)), dim3(256), 0, st, (int64_t)1, MM, mat.Kuu, MM, Lm, MM);
        if (jitter != 0.0) hipLaunchKernelGGL(add_diag_kernel, dim3(gridn(M)), dim3(256), 0, st, M, Lm, jitter);
        rc = 0;
    } else {
        rc = mxf_gram(h, kind, MXF_F64, 1, M, M, Q, Zd, 0, nullptr, 0, lsd, ard, 0, vard, 0, nullptr, 0, jitter, MXF_WRITE, Lm, M, MM, st);   // Kuu (+jitter) :69-72
    }
    if (rc) return rc;
    MXF_HIP(h, hipEventRecord(h->ev_fork, st));
    hipStream_t s2_ = h->side2;
    MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_fork, 0));
    MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_fork, 0));
    // second side stream, first thing: Su (H0 on the critical path needs it; its Cholesky comes later and is off the critical path)
    MXF_HIP(h, hipMemsetAsync(sc, 0, 16 * sizeof(D), s2_));
    MXF_HIP(h, hipMemsetAsync(scal, 0, 2 * (size_t)S * sizeof(D), s2_));
    // (r04) the accumulate-into outputs of the fused reverse pass and of the core's reverse mode are cleared HERE, on the second side stream at
    // the start of the call (it joins the caller's stream before the reverse pass): nine fills that used to sit between the T product and the
    // reverse pass on the critical path (~10 us of queue latency apiece: 0.1 ms of a 4.6 ms per-rank step)
    const bool early_clear = want_grad && !het;
    if (early_clear) {
        if (dY) MXF_HIP(h, hipMemsetAsync(dY, 0, sizeof(T) * (size_t)(sY == 0 ? B : SB) * P, s2_));
        if (dZ) MXF_HIP(h, hipMemsetAsync(dZ, 0, sizeof(T) * M * Q, s2_));
        if (dls) MXF_HIP(h, hipMemsetAsync(dls, 0, sizeof(T) * lsn, s2_));
        if (dvar) MXF_HIP(h, hipMemsetAsync(dvar, 0, sizeof(T), s2_));
        if (dX) MXF_HIP(h, hipMemsetAsync(dX, 0, sizeof(T) * (size_t)SB * Q, s2_));
        MXF_HIP(h, hipMemsetAsync(R, 0, sizeof(T) * MP, s2_));
        if (!use_mat) {
            MXF_HIP(h, hipMemsetAsync(dZc, 0, sizeof(D) * M * Q, s2_));
            MXF_HIP(h, hipMemsetAsync(dlsc, 0, sizeof(D) * lsn, s2_));
            MXF_HIP(h, hipMemsetAsync(dvc, 0, sizeof(D) * 4, s2_));
        }
    }
    if (!het && !het_stream) hipLaunchKernelGGL((convert_kernel<T, D>), dim3(1), dim3(256), 0, s2_, (int64_t)1, (int64_t)1, noise, (int64_t)1, noised, (int64_t)1);
    hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(MP)), dim3(256), 0, s2_, (int64_t)1, (int64_t)MP, mu, (int64_t)MP, mud, (int64_t)MP);
    hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(MM)), dim3(256), 0, s2_, (int64_t)1, (int64_t)MM, W, (int64_t)MM, Wd, (int64_t)MM);
    hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(M)), dim3(256), 0, s2_, (int64_t)1, (int64_t)M, sdiag, (int64_t)M, sd, (int64_t)M);
    hipLaunchKernelGGL((diag_embed_kernel<D>), dim3(gridn(MM)), dim3(256), 0, s2_, M, (const D*)sd, Su);
    rc = mxf_gemm_internal(h, MXF_F64, 0, 1, M, M, M, 1.0, Wd, M, 0, Wd, M, 0, 1.0, Su, M, 0, 1, 0, s2_);       // Su = W W^T + diag(s) :76
    if (rc) return rc;
    MXF_HIP(h, hipEventRecord(h->ev_su, s2_));           // H0 needs Su only; its Cholesky (log-det, Su^-1 for the reverse mode) is OFF the critical path
    MXF_STAGE(h, "Su formed (s2)", s2_);
    // Enqueue order = priority order (the host needs ~5 us per launch and a step has ~280 of them): first the few launches that carry
    // the bulk of the device work (Grams, Psi2), then the latency-critical Kuu chain, then the Su chain.
    // ---- side stream: Kuf_all, Kfu_all, Psi2 --------------------------------------------------------------------------------
    if (use_mat) {
        // (r06: the caller's Gram is only ever read -- no copy into the scratch (268 MB, 0.13 ms at 512 x 131 072))
        Kuf = const_cast<T*>(mat.Kuf);
    } else if (whiten) {
        // whitened tier: only the Kfu planes (operand (n, k = m) of V = L^-1 Kuf); they need nothing from the core, so they are written
        // first; V, its transposition (+ U = a^T V) and Phi = V V^T follow on this stream once L^-1 exists (below, after the Kuu chain
        // has been queued)
        // (r05 probe knob MXF_SVGP_WH_DEFER = 1 / 2: the pass held back until potrf(Kuu) / trtri(Kuu) has finished -- in this form the Kuu chain is a
        //  serial prefix of everything, and next to the planes pass its float64 workgroups wait for CUs: a 60 us GEMM inside potrf took 1.26 ms)
        static const int wh_defer = (int)MXF_KNOB("MXF_SVGP_WH_DEFER", 0);
        if (!wh_defer) {
        MXF_T0(h, MXF_T_PLANES_A, sd_);
        rc = mxf_gram_planes_internal(h, kind, SB, M, Q, (const float*)X, (const float*)Z, (const float*)ls, ard, (const float*)var, plKfu,
                                      (int64_t)pl_big, gscr1, sd_, split_mode);
        if (rc) return rc;
        MXF_T1(h, MXF_T_PLANES_A, sd_);
        MXF_STAGE(h, "Kfu planes (sd)", sd_);
        }
    } else if (use_split) {
        // float32 training step: the Grams are written directly as split planes (two scaled f16 terms = 4 bytes per element, never as f32):
        // Kuf planes (operand (m, k = n)) feed Psi2 and come first so that Psi2 (MFMA bound) starts early; the Kfu planes (operand
        // (n, k = m): T GEMM and the w^T Kuf row) are then written (HBM bound) on the second side stream NEXT TO Psi2.
        MXF_T0(h, MXF_T_PLANES_A, sd_);
        rc = mxf_gram_planes_internal(h, kind, M, SB, Q, (const float*)Z, (const float*)X, (const float*)ls, ard, (const float*)var, plKuf,
                                      (int64_t)pl_big, gscr0, sd_, split_mode, nullptr, 0, nullptr, 0, het_stream ? (const float*)hcs : nullptr, nullptr, B);
        if (rc) return rc;
        MXF_T1(h, MXF_T_PLANES_A, sd_);
        MXF_STAGE(h, "Kuf planes (sd)", sd_);
        if (bt_path) MXF_HIP(h, hipEventRecord(h->ev_aux, sd_));     // the T product reads THESE planes
        // (the Kfu planes -- operand (n, k = m) of the T GEMM -- are written later, on the second side stream, once w = Kuu^-1 mu exists:
        //  the same pass then also forms the row U = w^T Kuf)
    } else {
        rc = mxf_gram(h, kind, dtype, 1, M, SB, Q, Z, 0, X, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, Kuf, SB, 0, sd_);          // Kuf_all = k(Z, X_all) :73
        if (rc) return rc;
    }
    if (!use_split) MXF_HIP(h, hipEventRecord(h->ev_aux, sd_));          // Kuf ready: the T GEMM waits for it
    if (want_grad && !het && !whiten) {
        // Psi2 = Kuf Kuf^T depends on neither the core nor the T GEMM nor the reverse pass: it starts at once on the side stream,
        // lower blocks only, split-K.  float32: from the split planes of Kuf (gemm_split.hip); float64 / fallback: from the TRANSPOSED
        // Gram Kfu (S*B x M, rows = contiguous lines) as a TN GEMM (the NT form on Kuf reads 256 K-strided streams per workgroup).
        // Two launches: phase A covers the first KA = 128 M columns (about as long as the core chains run) with ONE workgroup per CU on
        // ~216 CUs, so that the core chains' f64 workgroups (a whole CU's LDS / registers each) still find free CUs; phase B (the rest)
        // fills the chip.  Same-box A/B at 4 samples per GPU: 13.65 -> 12.95 ms per step; neutral at 32 samples.
        // (few samples per GPU: the whole product runs in the reduced-occupancy form -- the core chains are the critical path there and
        //  Psi2 is short; same-box at 4 samples: 5.46 -> 5.20 ms per step; at 32 samples a longer first phase costs 0.1-0.4 ms)
        const int64_t ka_dflt = (use_split ? 192 : 128) * M;
        // r06, one set of planes (bt_path): nothing but the core chains runs next to Psi2 any more (the second planes pass used to take the
        // CUs Psi2 left free and starve the chain behind it), so the whole product leaves them 32 CUs and there is no reduced first phase --
        // the chains finish under Psi2 and T starts when Psi2 ends (same box, 32 samples: 23.5 -> 22.85 ms per step; 24 CUs: 23.8; 40: 22.9)
        const int64_t ka_req = psi2_ka >= 0 ? psi2_ka : (use_split && SB <= 2 * ka_dflt ? SB : (bt_path ? 0 : ka_dflt));
        const int rb_phase_b = (bt_path && !MXF_KNOB_SET("MXF_SVGP_PSI2_RB")) ? 32 : psi2_rb;
        const int64_t KA = (ka_req > 0 && ka_req < SB) ? ka_req / 32 * 32 : (ka_req > 0 ? SB : 0);
        MXF_T0(h, MXF_T_PSI2, sd_);
        if (use_split) {
            if (KA > 0) {
                // 3 workgroups of the three-term kernel fit a CU: (256 - 184) * 3 = 216 workgroups = one per CU on 216 CUs; the two-term
                // kernel fits 4: (256 - 202) * 4 = 216.  r06, one-planes path: (256 - 214) * 4 = 168 workgroups -- with the second planes pass
                // gone the chains are alone on what Psi2 leaves, and 88 CUs serve the few-sample step better than 40 (same box, 4 samples:
                // 4.30 -> 4.21 ms; 210: 4.25-4.32, 218: 4.22-4.26, 224: 4.27; configs[3] at 4 samples 2.41 -> 2.37)
                rc = mxf_gemm_split_internal(h, M, M, KA, (double)split_ga * split_ga, plKuf, (int64_t)pl_big, plKuf, (int64_t)pl_big, 0.0, (float*)Psi2, M, 1, sd_,
                                             psi2_ra_env ? psi2_ra : (split_mode == MXF_SPLIT_F16X2 ? (bt_path ? 214 : 202) : 184), split_mode, split_var, 2, nullptr);
                if (rc) return rc;
            }
            if (KA < SB) {
                const unsigned short* pk = plKuf + (KA / 16) * M * 16;
                rc = mxf_gemm_split_internal(h, M, M, SB - KA, (double)split_ga * split_ga, pk, (int64_t)pl_big, pk, (int64_t)pl_big, KA > 0 ? 1.0 : 0.0, (float*)Psi2, M, 1, sd_,
                                             rb_phase_b, split_mode, split_var, 2, nullptr);
                if (rc) return rc;
            }
        } else {
            rc = mxf_gram(h, kind, dtype, 1, SB, M, Q, X, 0, Z, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, Kfu, M, 0, sd_);
            if (rc) return rc;
            if (KA > 0) {
                rc = mxf_gemm_internal(h, dtype, 1, 0, M, M, KA, 1.0, Kfu, M, 0, Kfu, M, 0, 0.0, Psi2, M, 0, 1, 1, sd_, psi2_ra);
                if (rc) return rc;
            }
            if (KA < SB) {
                rc = mxf_gemm_internal(h, dtype, 1, 0, M, M, SB - KA, 1.0, Kfu + KA * M, M, 0, Kfu + KA * M, M, 0, KA > 0 ? 1.0 : 0.0, Psi2, M, 0, 1, 1,
                                       sd_, psi2_rb);
                if (rc) return rc;
            }
        }
        hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, sd_, Psi2, M, M, MM);
        MXF_T1(h, MXF_T_PSI2, sd_);
        MXF_STAGE(h, "Psi2 (sd)", sd_);
        MXF_HIP(h, hipEventRecord(h->ev_join2, sd_));
    }
    // ---- main stream: Kuu -> L -> L^-1 -> Ki, w (the critical path up to the T GEMM) --------------------------------------------
    // condition number of Kuu + jitter I (1-norm), for the float32 validity check of mxf_svgp_last_cond: |Kuu|_1 here, |Ki|_1 below
    // r06 probe knob MXF_SVGP_CHAIN_LATE (one-planes path, many samples): the Kuu chain (2: the Su chain too) starts only when the planes pass is
    // done.  A profiler timeline shows the planes pass at 2.17 instead of 1.7 ms next to the chains -- without the profiler it takes 1.79 ms
    // either way and the step does not move (same box: 22.60-22.69 / 22.66-22.77 / 22.60-22.70 ms for 0 / 1 / 2).  Off.
    static const int chain_late = (int)MXF_KNOB("MXF_SVGP_CHAIN_LATE", 0);
    const bool late = chain_late && bt_path && want_grad && SB > 2 * 192 * M;
    if (late) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_aux, 0));
    // r06 (MXF_SVGP_OFFPATH, default on): what the T product does not need leaves the caller's stream -- the two condition norms (64 workgroups walking
    // 16 rows each: 0.05 ms alone, 0.22 ms in front of the factorisation while the planes pass holds every CU), mu.w, tr(Ki Su), log|Kuu| go to
    // the second side stream; w = Ki mu is a one-wave-per-row kernel.  Kuu is copied for its norm (the factorisation works in place).
    static const int offpath_env = (int)MXF_KNOB("MXF_SVGP_OFFPATH", 1);
    const bool offpath = offpath_env != 0;
    if (offpath) {
        hipLaunchKernelGGL((convert_kernel<D, D>), dim3(gridn(MM)), dim3(256), 0, st, (int64_t)1, MM, (const D*)Lm, MM, Sui, MM);     // (Su^-1's buffer: written at the end of the Su chain, on the same stream as the norm)
        MXF_HIP(h, hipEventRecord(h->ev_k1, st));
        MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_k1, 0));
        hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s2_, M, (const double*)Sui, M, h->cond_dev);
    } else
    hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, M, (const double*)Lm, M, h->cond_dev);
    rc = mxf_potrf_internal(h, MXF_F64, 1, M, Lm, M, MM, info, st, false, false);                     // L :83 (trtri / sumlogdiag read the lower triangle only)
    if (rc) return rc;
    MXF_STAGE(h, "potrf Kuu", st);
    static const int wh_defer2 = (int)MXF_KNOB("MXF_SVGP_WH_DEFER", 0);
    auto deferred_planes = [&]() -> int {        // (probe knob, see above)
        MXF_HIP(h, hipEventRecord(h->ev_join2, st));
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_join2, 0));
        MXF_T0(h, MXF_T_PLANES_A, sd_);
        int r_ = mxf_gram_planes_internal(h, kind, SB, M, Q, (const float*)X, (const float*)Z, (const float*)ls, ard, (const float*)var, plKfu,
                                          (int64_t)pl_big, gscr1, sd_, split_mode);
        MXF_T1(h, MXF_T_PLANES_A, sd_);
        return r_;
    };
    if (whiten && wh_defer2 == 1) { rc = deferred_planes(); if (rc) return rc; }
    rc = mxf_trtri_internal(h, MXF_F64, 1, M, Lm, M, MM, Linv, M, MM, st);
    if (rc) return rc;
    MXF_STAGE(h, "trtri Kuu", st);
    if (whiten && wh_defer2 == 2) { rc = deferred_planes(); if (rc) return rc; }
    static const int phi_late_env = (int)MXF_KNOB("MXF_SVGP_WH_PHI_LATE", 0);
    const bool phi_late = phi_late_env && whiten && bt_wh && want_grad && !het && SB <= 2 * 192 * M;
    auto launch_phi = [&](bool few_) -> int {
        MXF_T0(h, MXF_T_PSI2, sd_);
        int r_ = mxf_gemm_split_internal(h, M, M, SB, (double)split_ga * split_ga, plKuf, (int64_t)pl_big, plKuf, (int64_t)pl_big, 0.0, (float*)Psi2, M, 1, sd_,
                                         few_ ? 202 : psi2_rb, split_mode, split_var, 1, nullptr);
        if (r_) return r_;
        hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, sd_, Psi2, M, M, MM);
        MXF_T1(h, MXF_T_PSI2, sd_);
        MXF_STAGE(h, "Phi (sd)", sd_);
        return 0;
    };
    unsigned* limax = (unsigned*)(info2 + 4);           // bit pattern of max |L^-1| (whitened tier; word cleared by svgp_init_kernel)
    if (whiten) {
        // L^-1 as f16x2 planes (the A operand of V = L^-1 Kuf); Aext is free until Hh is formed
        hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, st, M, M, (const D*)Linv, M, Aext, M);
        rc = mxf_maxabs_internal(h, M, M, (const float*)Aext, M, limax, st, false);
        if (rc) return rc;
        rc = mxf_split_planes_internal(h, M, M, (const float*)Aext, M, plLi, st, MXF_SPLIT_F16X2, limax);
        if (rc) return rc;
        // a = L^-1 mu (float64, then the streaming dtype): the V product's epilogue forms U = a^T V from it
        MXF_HIP(h, hipStreamWaitEvent(st, h->ev_su, 0));                                              // mu (second side stream)
        if (offpath) hipLaunchKernelGGL((trmv_lower_kernel<D>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, M, P, (const D*)Linv, M, (const D*)mud, (int64_t)P, ad);
        else {
        rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, P, M, 1.0, Linv, M, 0, mud, P, 0, 0.0, ad, P, 0, 1, 0, st);
        if (rc) return rc;
        }
        hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MP)), dim3(256), 0, st, (int64_t)1, MP, (const D*)ad, MP, aT, MP);
        MXF_HIP(h, hipEventRecord(h->ev_aux2, st));                                                   // L^-1 planes and a ready
        // side stream: V = L^-1 Kuf, written as the planes of the (m, k = n) operand holding V / sigma * 2^14 (|v_n|^2 <= k_nn = sigma^2):
        // acc = (s_L L^-1) (Kfu / sigma^2 2^14)^T  ->  acc sigma / s_L.  L^-1 is lower triangular: a row tile's k loop stops at its last row.
        // The SAME launch writes the planes of V^T (the (n, k = m) operand of T = Hh V) and, for one output column, the 128-row partial sums
        // of U = a^T V (r04 late: a separate transposition + U pass over the planes took 3.3 ms of the 35 ms step -- 17 GB of traffic).
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_aux2, 0));
        MXF_T0(h, MXF_T_VGEMM, sd_);
        // (few samples per GPU: the Kuu chain on the caller's stream is the critical path -- both side-stream products leave it 40 CUs)
        const bool few = SB <= 2 * 192 * M;
        if (bt_wh)      // planes of V only: T and U come from them (gemm_bt.hip)
            rc = mxf_gemm_split_internal(h, M, SB, M, 1.0, plLi, (int64_t)pl_h0, plKfu, (int64_t)pl_big, 0.0, nullptr, SB, 0, sd_, few ? 40 : 0, split_mode, sigf, 1,
                                         (const unsigned*)limax, nullptr, 0, nullptr, plKuf, (int64_t)pl_big, 1);
        else
        rc = mxf_gemm_split_internal(h, M, SB, M, 1.0, plLi, (int64_t)pl_h0, plKfu, (int64_t)pl_big, 0.0, nullptr, SB, 0, sd_, few ? 40 : 0, split_mode, sigf, 1,
                                     (const unsigned*)limax, nullptr, 0, nullptr, plKuf, (int64_t)pl_big, 1, plVt, pVt, P == 1 ? (const float*)aT : nullptr,
                                     P == 1 ? upart : nullptr);
        if (rc) return rc;
        MXF_T1(h, MXF_T_VGEMM, sd_);
        MXF_STAGE(h, "V = Linv Kuf (sd)", sd_);
        MXF_T0(h, MXF_T_PLANES_B, sd_);
        if (bt_wh) rc = 0;
        else if (P == 1) rc = mxf_upart_reduce_internal(h, SB, (int)(M / 128), upart, sigf, 1.f / 16384.f, (float*)(Text + M * SB), sd_);
        else hipLaunchKernelGGL((wt_planes_kernel<8, 2>), dim3((unsigned)((SB + 255) / 256)), dim3(256), 0, sd_, M, SB, P, (const unsigned short*)plVt,
                                pVt, (const float*)aT, (float*)(Text + M * SB), (const float*)sigf);
        if (rc) return rc;
        MXF_T1(h, MXF_T_PLANES_B, sd_);
        MXF_HIP(h, hipEventRecord(h->ev_aux, sd_));      // V^T planes and U ready: the T GEMM / the reverse pass wait for it
        // Phi = V V^T (lower tiles, split-K) = sigma^2 2^-28 (planes)(planes)^T
        // r06 probe knob MXF_SVGP_WH_PHI_LATE, few samples per GPU: Phi is enqueued BEHIND the T product (launch_phi above) -- T is on the
        // critical path (the reverse pass needs it), Phi only feeds the core's reverse mode in the tail.  Measured, same box, 4 samples at
        // trained-like parameters: 6.14-6.17 ms with it against 6.04 -- Phi then runs next to the latency-bound reverse pass and the tail waits
        // for it; whitened parity tests pass either way.  Off.
        if (!phi_late) { rc = launch_phi(few); if (rc) return rc; }
    }
    // r06 (MXF_SVGP_SYM_LOWER): the symmetric products of the chain -- Ki here, H0 below -- as lower tiles + mirror (half the work of a product that
    // runs on the 88 CUs Psi2 leaves in the few-sample regime; the results become exactly symmetric)
    static const int symlow_env = (int)MXF_KNOB("MXF_SVGP_SYM_LOWER", 1);
    // (float32 mode only: the float64 path is the parity path and stays operation for operation what the trajectory tests were recorded with --
    //  tests/test_svgp_notebook.py's 100-epoch float64 run moved its learned noise by 12 % with exactly symmetric Ki / H0, past its 10 % band)
    const int symlow = (symlow_env && sizeof(T) == 4) ? 1 : 0;
    rc = mxf_gemm_internal(h, MXF_F64, 1, 0, M, M, M, 1.0, Linv, M, 0, Linv, M, 0, 0.0, Ki, M, 0, 1, symlow, st);   // Ki = Linv^T Linv
    if (rc) return rc;
    if (symlow) hipLaunchKernelGGL((symmetrize_kernel<D>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, st, Ki, M, M, MM);
    if (!offpath) hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, M, (const double*)Ki, M, h->cond_dev + 1);
    MXF_HIP(h, hipStreamWaitEvent(st, h->ev_su, 0));                                                  // Su, mu, noise, accumulators (second side stream)
    if (offpath) hipLaunchKernelGGL((gemv_rows_kernel<D>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, M, P, (const D*)Ki, M, (const D*)mud, (int64_t)P, wd);   // w = Ki mu
    else {
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, P, M, 1.0, Ki, M, 0, mud, P, 0, 0.0, wd, P, 0, 1, 0, st);       // w = Ki mu
    if (rc) return rc;
    }
    if (!offpath) hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MP)), dim3(256), 0, st, MP, (const D*)mud, (const D*)wd, 1.0, sc + 3);
    hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MP)), dim3(256), 0, st, (int64_t)1, MP, (const D*)wd, MP, wT, MP);      // w in the streaming dtype
    MXF_STAGE(h, "Ki, w", st);
    if (use_split && !whiten) MXF_HIP(h, hipEventRecord(h->ev_aux2, st));                             // w ready: the Kfu planes + U pass may start
    if (!offpath) { rc = mxf_sumlogdiag_internal(h, MXF_F64, 1, M, Lm, M, MM, sc + 0, st); if (rc) return rc; }
    // ---- second side stream: Su -> Ls -> Su^-1 ----------------------------------------------------------------------------------
    // (a plain kernel, not hipMemcpyAsync: the runtime's copy path sat idle for ~1 ms before it started next to busy queues -- r02 timeline:
    //  the Su chain did not begin until 1.66 ms although nothing in the Kuu chain feeds it)
    if (late && chain_late >= 2) MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_aux, 0));
    hipLaunchKernelGGL((convert_kernel<D, D>), dim3(gridn(MM)), dim3(256), 0, s2_, (int64_t)1, MM, (const D*)Su, MM, tmp, MM);
    rc = mxf_potrf_internal(h, MXF_F64, 1, M, tmp, M, MM, info2, s2_, false, false);                  // Ls = chol(Su) :84
    if (rc) return rc;
    MXF_STAGE(h, "potrf Su (s2)", s2_);
    rc = mxf_sumlogdiag_internal(h, MXF_F64, 1, M, tmp, M, MM, sc + 1, s2_);
    if (rc) return rc;
    if (want_grad) {
        rc = mxf_trtri_internal(h, MXF_F64, 1, M, tmp, M, MM, Lsinv, M, MM, s2_);
        if (rc) return rc;
        rc = mxf_gemm_internal(h, MXF_F64, 1, 0, M, M, M, 1.0, Lsinv, M, 0, Lsinv, M, 0, 0.0, Sui, M, 0, 1, 0, s2_);
        if (rc) return rc;
    }
    MXF_STAGE(h, "Su^-1 (s2)", s2_);
    MXF_HIP(h, hipEventRecord(h->ev_join, s2_));
    if (use_split && !whiten && !bt_path) {
        // Kfu planes (operand (n, k = m) of the T GEMM) + the row U = w^T Kuf in ONE pass, behind the Su chain on the second side stream:
        // HBM-write bound.  (r03, tests/probes/svgp_stages.py: the pass starts when w = Kuu^-1 mu exists, and the Kuu chain -- potrf, trtri,
        // Ki -- shares the chip with the Kuf planes pass and Psi2 and finishes just after Psi2: 0.9 / 1.4 / 1.6 ms at 4 samples against 0.6
        // alone.  A stream of its own for this pass changes nothing: H0, hence the T GEMM, waits for the same chain.)
        MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_aux2, 0));
        MXF_T0(h, MXF_T_PLANES_B, s2_);
        rc = mxf_gram_planes_internal(h, kind, SB, M, Q, (const float*)X, (const float*)Z, (const float*)ls, ard, (const float*)var, plKfu,
                                      (int64_t)pl_big, gscr1, s2_, split_mode, (const float*)wT, P, (float*)(Text + M * SB), SB, nullptr,
                                      het_stream ? (const float*)hrs : nullptr, B);
        if (rc) return rc;
        MXF_T1(h, MXF_T_PLANES_B, s2_);
        MXF_STAGE(h, "Kfu planes + U (s2)", s2_);
        MXF_HIP(h, hipEventRecord(h->ev_aux, s2_));      // Kfu planes and U ready: the T GEMM / the reverse pass wait for it
    }
    MXF_HIP(h, hipStreamWaitEvent(st, h->ev_su, 0));                                                  // Su formed (second side stream)
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Ki, M, 0, Su, M, 0, 0.0, KiSu, M, 0, 1, 0, st);
    if (rc) return rc;
    if (whiten) {
        // Hh = L^-T (I - A_s A_s^T) = L^-T - Ki Su L^-T   (T = Hh V; |Hh| ~ sqrt(cond) where |H0| ~ cond)
        hipLaunchKernelGGL((transpose_convert_kernel<D, D>), dim3(gridn(MM)), dim3(256), 0, st, M, M, (const D*)Linv, M, H0, M);
        rc = mxf_gemm_internal(h, MXF_F64, 0, 1, M, M, M, -1.0, KiSu, M, 0, Linv, M, 0, 1.0, H0, M, 0, 1, 0, st);
        if (rc) return rc;
    } else {
    hipLaunchKernelGGL((convert_kernel<D, D>), dim3(gridn(MM)), dim3(256), 0, st, (int64_t)1, MM, (const D*)Ki, MM, H0, MM);     // H0 <- Ki (a plain kernel: the runtime's copy engine path costs ~10x as much next to busy queues)
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, -1.0, KiSu, M, 0, Ki, M, 0, 1.0, H0, M, 0, 1, symlow, st);     // H0 = Ki - Ki Su Ki
    if (rc) return rc;
    if (symlow) hipLaunchKernelGGL((symmetrize_kernel<D>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, st, H0, M, M, MM);
    }
    if (!offpath) hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MM)), dim3(256), 0, st, MM, (const D*)Ki, (const D*)Su, 1.0, sc + 2);
    // A_ext = [H0 ; w^T] in the streaming dtype
    // (the w^T row that used to follow H0 in A_ext was read by nothing since U = w^T Kuf left the product: its launch is gone)
    const bool fused_h0max = use_split && split_mode == MXF_SPLIT_F16X2 && sizeof(T) == 4;
    if (fused_h0max) hipLaunchKernelGGL(convert_max_kernel, dim3((unsigned)(gridn(MM) > 256 ? 256 : gridn(MM))), dim3(256), 0, st, MM, (const D*)H0, (float*)Aext, (unsigned*)(info2 + 2));
    else
    hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, st, M, M, (const D*)H0, M, Aext, M);

    // ---- streaming part -----------------------------------------------------------------------------------
    // [T; U] = [H0; w^T] Kuf_all
    // (training step on the split path whose reverse pass runs on the matrix pipe: T is written in 16-column blocks, so that each 16 x 16
    //  tile that pass reads is one contiguous KB instead of 16 pieces of 64 bytes, 4 SB bytes apart)
    const int t_blocked = (use_split && want_grad && !het && SB % 16 == 0 && mxf_svgp_bwd_reads_blocked(kind, dtype, SB, B, Q, P, Text)) ? 1 : 0;
    if (use_split) {
        unsigned* h0max = (unsigned*)(info2 + 2);       // bit pattern of max |H0|: the power-of-two scale of its f16x2 planes
        if (split_mode == MXF_SPLIT_F16X2 && !fused_h0max) { rc = mxf_maxabs_internal(h, M, M, (const float*)Aext, M, h0max, st, false); if (rc) return rc; }      // (word cleared by svgp_init_kernel)
        rc = mxf_split_planes_internal(h, M, M, (const float*)Aext, M, plH0, st, split_mode, split_mode == MXF_SPLIT_F16X2 ? h0max : nullptr);
        if (rc) return rc;
    }
    MXF_T1(h, MXF_T_CHAIN, st);
    MXF_STAGE(h, "H0 planes", st);
    MXF_HIP(h, hipEventRecord(h->ev_fork, st));                                                       // core (Ki, KiSu, H0, w) ready
    if (offpath) {      // |Ki|_1, mu.w, tr(Ki Su), log|Kuu| behind the Su chain on the second side stream; the caller's stream picks them up behind the T product
        MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_fork, 0));
        hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s2_, M, (const double*)Ki, M, h->cond_dev + 1);
        hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MP)), dim3(256), 0, s2_, MP, (const D*)mud, (const D*)wd, 1.0, sc + 3);
        hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MM)), dim3(256), 0, s2_, MM, (const D*)Ki, (const D*)Su, 1.0, sc + 2);
        rc = mxf_sumlogdiag_internal(h, MXF_F64, 1, M, Lm, M, MM, sc + 0, s2_);
        if (rc) return rc;
        MXF_HIP(h, hipEventRecord(h->ev_k3, s2_));
    }
    MXF_HIP(h, hipStreamWaitEvent(st, h->ev_aux, 0));                                                 // Kuf_all from the side stream
    // (in-step timing only: the T product is held until Psi2 has finished, so that `t_gemm` is the product's own duration -- untimed, its
    //  first workgroups take the CUs Psi2's last work items free, and the event pair would count that queueing)
    if (h->tm.on && bt_path && want_grad && !het) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_join2, 0));
    MXF_T0(h, MXF_T_TGEMM, st);
    // r05: the reverse pass as the EPILOGUE of the T product (gemm_split.hip wide_body<..., FUSE>): T is never written
    const bool fuse_bwd = use_split && want_grad && !het && !het_stream && split_mode == MXF_SPLIT_F16X2 && !bt_path && !bt_wh &&
                          mxf_svgp_bwd_fuse_ok(kind, dtype, M, SB, B, Q, P) != 0;
    mxf_fuse_args fza;
    if (fuse_bwd) {
        rc = mxf_svgp_bwd_fuse_prepare(h, M, SB, B, Q, (const float*)Z, (const float*)X, (const float*)ls, ard, (const float*)var,
                                       (const float*)(Text + M * SB), (const float*)Y, sY, (const float*)wT, (const float*)noise, a1, (float*)dX, (float*)dY,
                                       (sY == 0 && SS > 1) ? 1 : 0, scal, (const unsigned*)(info2 + 2), &fza, st);
        if (rc) return rc;
    }
    if (fuse_bwd && whiten)
        rc = mxf_gemm_split_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plVt, pVt, 0.0, (float*)Text, SB, 0, st, 0, split_mode,
                                     sigf, 1, (const unsigned*)(info2 + 2), nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, &fza);
    else if (fuse_bwd)
        rc = mxf_gemm_split_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plKfu, (int64_t)pl_big, 0.0, (float*)Text, SB, 0, st, 0, split_mode,
                                     split_var, 1, (const unsigned*)(info2 + 2), nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, &fza);
    else if (bt_wh)       // T = Hh V from the planes of V (scaled from max |Hh|; V / sigma 2^14), U = a^T V from the same fragments
        rc = mxf_gemm_bt_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plKuf, (int64_t)pl_big, M, (float*)Text, SB, t_blocked, st, 0,
                                  sigf, (const unsigned*)(info2 + 2), (unsigned*)(info2 + 3), (const float*)aT, (float*)(Text + M * SB),
                                  1.0 / 16384.0, wpl);
    else if (whiten)      // T = Hh V: planes of Hh (scaled from max |Hh|) x planes of V^T (V / sigma 2^14)
        rc = mxf_gemm_split_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plVt, pVt, 0.0, (float*)Text, SB, 0, st, 0, split_mode,
                                     sigf, 1, (const unsigned*)(info2 + 2), nullptr, t_blocked, (unsigned*)(info2 + 3));
    else if (bt_path)     // T = H0 Kuf straight from the Kuf planes, U = w^T Kuf from the same fragments (gemm_bt.hip)
        rc = mxf_gemm_bt_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plKuf, (int64_t)pl_big, M, (float*)Text, SB, t_blocked, st, 0,
                                  split_var, (const unsigned*)(info2 + 2), (unsigned*)(info2 + 3), (const float*)wT, (float*)(Text + M * SB),
                                  1.0 / 16384.0, wpl);
    else if (use_split)   // T = H0 Kuf = H0 Kfu^T on the 16-bit matrix pipe (f32-equivalent splitting, gemm_split.hip)
        rc = mxf_gemm_split_internal(h, M, SB, M, (double)split_ga, plH0, (int64_t)pl_h0, plKfu, (int64_t)pl_big, 0.0, (float*)Text, SB, 0, st, 0, split_mode,
                                     split_var, 1, split_mode == MXF_SPLIT_F16X2 ? (const unsigned*)(info2 + 2) : nullptr, nullptr, t_blocked,
                                     (unsigned*)(info2 + 3));       // max |T| for the reverse pass (word cleared by svgp_init_kernel)
    else if (het_split) {
        rc = mxf_maxabs_internal(h, M, M, (const float*)Aext, M, hsw + 0, st);
        if (!rc) rc = mxf_split_planes_internal(h, M, M, (const float*)Aext, M, hsA, st, MXF_SPLIT_F16X2, hsw + 0);
        if (!rc) rc = mxf_maxabs_internal(h, M, SB, (const float*)Kuf, SB, hsw + 1, st);
        if (!rc) rc = mxf_split_planes_internal(h, M, SB, (const float*)Kuf, SB, hsK, st, MXF_SPLIT_F16X2, hsw + 1);      // (m, k = n): the K-major operand of T, the row operand of G'
        if (!rc) rc = mxf_gemm_bt_internal(h, M, SB, M, 1.0, hsA, (int64_t)pl_h0, hsK, (int64_t)pl_big, M, (float*)Text, SB, 0, st, 0, nullptr, hsw + 0, nullptr,
                                           nullptr, nullptr, 1.0, nullptr, hsw + 1);
    } else
        rc = mxf_gemm_internal(h, dtype, 0, 0, M, SB, M, 1.0, Aext, M, 0, Kuf, SB, 0, 0.0, Text, SB, 0, 1, 0, st);   // T = H0 Kuf (MFMA)
    if (rc) return rc;
    MXF_T1(h, MXF_T_TGEMM, st);
    MXF_STAGE(h, "T", st);
    if (phi_late) {      // Phi behind T: next to the reverse pass instead of next to T
        MXF_HIP(h, hipEventRecord(h->ev_tg, st));
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_tg, 0));
        rc = launch_phi(true);
        if (rc) return rc;
    }
    if (use_split) {
        // (U = w^T Kuf was written by the Kfu planes pass)
    } else {
        constexpr int VEC = Vec16<T>::n;
        dim3 gu((unsigned)((SB + 256 * VEC - 1) / (256 * VEC)));
#ifndef MXF_NO_WTSPLIT
        if (gu.x <= 256 && M >= 64) {            // few columns: split the rows as well (the kernel then adds into U)
            // (r06: up to one workgroup per CU -- 512 x 131 072, the deep GP's first layer, was 128 workgroups walking 512 dependent rows: 0.24 ms)
            int64_t ch = 1024 / gu.x; if (ch > M / 16) ch = M / 16; if (ch > 64) ch = 64;
            if (ch > 1) { gu.y = (unsigned)ch; MXF_HIP(h, hipMemsetAsync(Text + M * SB, 0, sizeof(T) * (size_t)P * SB, st)); }
        }
#endif
        if (P == 1) hipLaunchKernelGGL((wt_kuf_kernel<T, 1>), gu, dim3(256), 0, st, M, SB, P, (const T*)Kuf, (const T*)wT, Text + M * SB);
        else hipLaunchKernelGGL((wt_kuf_kernel<T, 8>), gu, dim3(256), 0, st, M, SB, P, (const T*)Kuf, (const T*)wT, Text + M * SB);   // U = w^T Kuf
    }
    // the part of the core reverse mode that depends on Psi2 only (not on R): dSu = -Ki G Ki + bP/2 (Su^-1 - Ki), dW = 2 dSu W,
    // dSdiag = diag(dSu), T1 = G Ki Su.  Streaming path: queued on the side stream right behind Psi2, so it runs under the T GEMM /
    // the reverse pass instead of in the step's tail.
    // (r06) float32 mode: the whole R-independent part of the core's reverse mode -- dKuu0 included -- on the side stream, from X = Ki G Ki (below).
    // float64 mode keeps the r05 flow: forming A_Ki = G - G Ki Su - ... FIRST and multiplying by Ki afterwards cancels before it amplifies; the X form
    // multiplies first and loses ~2 digits on ill-conditioned Kuu (uncertain-input toy of tests/test_gpu_config4.py, cond ~ 1e6: float64 lengthscale
    // gradient 8.6e-10 -> 5.8e-8 against the oracle, tests/probes/r06_early_kuu_accuracy.py) -- nothing next to float32 streaming (1e-5), but the
    // float64 path is the parity path.  Probe knob MXF_SVGP_EARLY_KUU: 0 = r05 flow everywhere, 2 = X form in float64 too.
    static const int early_kuu_env = (int)MXF_KNOB("MXF_SVGP_EARLY_KUU", 1);
    const bool early_kuu = early_kuu_env == 2 || (early_kuu_env != 0 && sizeof(T) == 4);
    auto su_reverse = [&](hipStream_t s_, bool with_t1) -> int {
        int r_ = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Ki, M, 0, G, M, 0, 0.0, tmp, M, 0, 1, 0, s_);            // T3 = Ki G
        if (r_) return r_;
        hipLaunchKernelGGL((axpby_kernel<D>), dim3(gridn(MM)), dim3(256), 0, s_, MM, 1.0, (const D*)Sui, -1.0, (const D*)Ki, dSu);   // Sui - Ki
        r_ = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, -1.0, tmp, M, 0, Ki, M, 0, 0.5 * bw * P, dSu, M, 0, 1, 0, s_);
        if (r_) return r_;
        if (dW) {
            r_ = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 2.0, dSu, M, 0, Wd, M, 0, 0.0, Lsinv, M, 0, 1, 0, s_);     // dW = 2 dSu W (Lsinv buffer is free)
            if (r_) return r_;
            hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, s_, MM, (T)1, (const D*)Lsinv, dW, 0);
        }
        if (dSdiag) hipLaunchKernelGGL((diag_extract_kernel<D, T>), dim3(gridn(M)), dim3(256), 0, s_, M, (const D*)dSu, M, dSdiag);
        if (with_t1) r_ = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, G, M, 0, KiSu, M, 0, 0.0, T1, M, 0, 1, 0, s_);           // T1 = G Ki Su
        return r_;
    };
    bool side_split = false;
    if (want_grad && !het && early_kuu) {
        // r06: the whole R-independent part of the core's reverse mode from X = Ki G Ki (G = c Psi2, c = P a1 beta / 2):
        //   dSu   = -X + bP/2 (Su^-1 - Ki)                                  (as before)
        //   dKuu0 = -Ki A0 Ki - bP/2 Ki,  A0 = G - T1 - T1^T - b (P/2 Su + 1/2 mu mu^T),  T1 = G Ki Su
        //         = -X + Y + Y^T + b (P/2 Ki Su Ki + 1/2 w w^T) - bP/2 Ki,  Y = X (Ki Su)^T,  Ki Su Ki = Ki - H0 (explicit form)
        // i.e. FOUR M^3 products (Ki G, . Ki, dSu W, X (Ki Su)^T) where the r05 flow took six (Ki G, . Ki, dSu W, G Ki Su, Ki A, . Ki), the last two
        // of them in the step's tail behind the reverse pass.  Whitened form: X = c L^-T Phi L^-1 directly (Phi = V V^T; G = c L Phi L^T is never
        // formed) + Ki Su Ki as a product: five instead of eight.  These products run on the CUs the T product leaves free (few samples: 40), so
        // their number is the length of the side stream: stage stamps at 4 samples, r05 flow: Su reverse ends 3.70 ms, reverse pass 3.54, end 3.97.
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_fork, 0));      // Ki, KiSu, H0 (main)
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_join, 0));      // Su^-1; `tmp` (chol(Su)) is free from here on
        D* Xb = whiten ? G : T2;
        static const int xlow_env = (int)MXF_KNOB("MXF_SVGP_X_LOWER", 1);
        const int xlow = xlow_env ? 1 : 0;
        if (whiten) {
            hipLaunchKernelGGL((scale_beta_kernel<T>), dim3(gridn(MM)), dim3(256), 0, sd_, MM, (const T*)Psi2, (const D*)noised, 0.5 * P * a1, T2);   // c Phi
            hipLaunchKernelGGL((trace_kernel<D>), dim3(1), dim3(256), 0, sd_, M, (const D*)T2, M, (int64_t)0, sc + 6);                                 // c tr(Phi)
            rc = mxf_gemm_internal(h, MXF_F64, 1, 0, M, M, M, 1.0, Linv, M, 0, T2, M, 0, 0.0, tmp, M, 0, 1, 0, sd_);       // L^-T (c Phi)
            if (rc) return rc;
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, tmp, M, 0, Linv, M, 0, 0.0, Xb, M, 0, 1, xlow, sd_);    // X = c L^-T Phi L^-1 (symmetric: lower tiles, mirrored)
            if (rc) return rc;
            if (xlow) hipLaunchKernelGGL((symmetrize_kernel<D>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, sd_, Xb, M, M, MM);
            hipLaunchKernelGGL(dot_t_kernel, dim3(dotgrid(MM)), dim3(256), 0, sd_, M, (const D*)Su, (const D*)Xb, sc + 7);   // tr(Su X): the accurate total of the q_n
        } else {
            hipLaunchKernelGGL((scale_beta_kernel<T>), dim3(gridn(MM)), dim3(256), 0, sd_, MM, (const T*)Psi2, (const D*)noised, 0.5 * P * a1, G);
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Ki, M, 0, G, M, 0, 0.0, tmp, M, 0, 1, 0, sd_);           // Ki G
            if (rc) return rc;
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, tmp, M, 0, Ki, M, 0, 0.0, Xb, M, 0, 1, xlow, sd_);       // X = Ki G Ki (symmetric: lower tiles, mirrored)
            if (rc) return rc;
            if (xlow) hipLaunchKernelGGL((symmetrize_kernel<D>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, sd_, Xb, M, M, MM);
        }
        // (probe knob, off: measured neutral to slower -- per-rank step 3.79-3.84 ms without, 3.85-3.90 with it, tests/probes/r06_side_split.sh: the two
        //  branches then share the CUs the reverse pass leaves, and the tail still waits for the later of them)
        static const int side_split_env = (int)MXF_KNOB("MXF_SVGP_SIDE_SPLIT", 0);
        side_split = side_split_env != 0;
        if (side_split) { MXF_HIP(h, hipEventRecord(h->ev_k1, sd_)); MXF_HIP(h, hipStreamWaitEvent(s2_, h->ev_k1, 0)); }      // X ready
        hipLaunchKernelGGL(dsu_kernel, dim3(gridn(MM)), dim3(256), 0, sd_, MM, (const D*)Xb, (const D*)Sui, (const D*)Ki, 0.5 * bw * P, dSu);
        if (dW) {
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 2.0, dSu, M, 0, Wd, M, 0, 0.0, Lsinv, M, 0, 1, 0, sd_);     // dW = 2 dSu W (Lsinv buffer is free)
            if (rc) return rc;
            hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, sd_, MM, (T)1, (const D*)Lsinv, dW, 0);
        }
        if (dSdiag) hipLaunchKernelGGL((diag_extract_kernel<D, T>), dim3(gridn(M)), dim3(256), 0, sd_, M, (const D*)dSu, M, dSdiag);
        // the dKuu0 branch (Y, Ki Su Ki, dKuu0) needs X only: it runs on the second side stream next to the dSu / dW branch (MXF_SVGP_SIDE_SPLIT) --
        // in the few-sample regime this stream ends after the reverse pass, and the step's tail waits for it
        hipStream_t sk_ = side_split ? s2_ : sd_;
        rc = mxf_gemm_internal(h, MXF_F64, 0, 1, M, M, M, 1.0, Xb, M, 0, KiSu, M, 0, 0.0, T1, M, 0, 1, 0, sk_);             // Y = X (Ki Su)^T
        if (rc) return rc;
        if (whiten) {
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, KiSu, M, 0, Ki, M, 0, 0.0, AKi, M, 0, 1, 0, sk_);        // Ki Su Ki (H0 holds Hh in this form)
            if (rc) return rc;
        }
        hipLaunchKernelGGL(dkuu0_kernel, dim3(gridn(MM)), dim3(256), 0, sk_, M, P, (const D*)Xb, (const D*)T1, (const D*)Ki, whiten ? (const D*)AKi : (const D*)nullptr,
                           (const D*)H0, (const D*)wd, bw, dKuu);
        if (side_split) MXF_HIP(h, hipEventRecord(h->ev_k2, s2_));
    } else if (want_grad && !het) {
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_fork, 0));      // Ki, KiSu (main)
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_join, 0));      // Su^-1; `tmp` (chol(Su)) is free from here on
        if (whiten) {
            // G = d/dH0 = P a1 beta / 2 Psi2 with Psi2 = L Phi L^T in float64: the core's reverse mode then forms Ki G Ki = c L^-T Phi L^-1 --
            // the float32 error of Phi is amplified by |L^-1|^2 ~ cond, that of a float32 Psi2 by |Ki|^2 ~ cond^2
            hipLaunchKernelGGL((scale_beta_kernel<T>), dim3(gridn(MM)), dim3(256), 0, sd_, MM, (const T*)Psi2, (const D*)noised, 0.5 * P * a1, T2);
            rc = mxf_tril_copy_internal(h, M, Lm, AKi, sd_);
            if (rc) return rc;
            rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, AKi, M, 0, T2, M, 0, 0.0, dKuu, M, 0, 1, 0, sd_);
            if (rc) return rc;
            rc = mxf_gemm_internal(h, MXF_F64, 0, 1, M, M, M, 1.0, dKuu, M, 0, AKi, M, 0, 0.0, G, M, 0, 1, 0, sd_);
            if (rc) return rc;
        } else
        hipLaunchKernelGGL((scale_beta_kernel<T>), dim3(gridn(MM)), dim3(256), 0, sd_, MM, (const T*)Psi2, (const D*)noised, 0.5 * P * a1, G);
        rc = su_reverse(sd_, true);
        if (rc) return rc;
        if (whiten) {      // c tr(Phi) and c tr(Su L^-T Phi L^-1) = tr((Ki Su) (Ki G)): the accurate total of the q_n (svgp_finalize_kernel)
            hipLaunchKernelGGL((trace_kernel<D>), dim3(1), dim3(256), 0, sd_, M, (const D*)T2, M, (int64_t)0, sc + 6);
            hipLaunchKernelGGL(dot_t_kernel, dim3(dotgrid(MM)), dim3(256), 0, sd_, M, (const D*)KiSu, (const D*)tmp, sc + 7);
        }
    }
    if (want_grad && !het) {
        MXF_STAGE(h, "Su reverse (sd)", sd_);
        MXF_HIP(h, hipEventRecord(h->ev_join2, sd_));            // Psi2, G, T1, dSu outputs (early_kuu: dKuu0 as well)
    }
    MXF_HIP(h, hipStreamWaitEvent(st, h->ev_join, 0));      // Su chain complete (log-det for the value, Su^-1 for the reverse mode); hidden under the T GEMM
    if (offpath) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_k3, 0));      // ... and the condition norms and value scalars behind it
    const int dY_shared = (sY == 0 && SS > 1) ? 1 : 0;
    D* dnz = nullptr; D* dvdir = nullptr;
    if (het) {
        if (want_grad) {
            dvdir = sc + 5;
            if (dY) MXF_HIP(h, hipMemsetAsync(dY, 0, sizeof(T) * (size_t)(ysamp ? (int64_t)S * B : (sY == 0 ? B : SB)) * P, st));
            if (dZ && !use_mat) MXF_HIP(h, hipMemsetAsync(dZ, 0, sizeof(T) * M * Q, st));
            if (dls && !use_mat) MXF_HIP(h, hipMemsetAsync(dls, 0, sizeof(T) * lsn, st));
            if (dvar && !use_mat) MXF_HIP(h, hipMemsetAsync(dvar, 0, sizeof(T), st));
            if (dX && !use_mat) MXF_HIP(h, hipMemsetAsync(dX, 0, sizeof(T) * (size_t)SB * Q, st));
            if (dnoise) MXF_HIP(h, hipMemsetAsync(dnoise, 0, sizeof(T) * (size_t)nrows * ncols, st));
        }
        T* Ksc = Kfu;   // the transposed-Gram slot is unused on this path
        MXF_HIP(h, hipMemsetAsync(qbuf, 0, sizeof(T) * (size_t)SB, st));
        hipLaunchKernelGGL((svgp_het_rows_kernel<T>), dim3((unsigned)((SB + 255) / 256), (unsigned)((M + HET_RCH - 1) / HET_RCH)), dim3(256), 0, st, SB, B, M,
                           P, SYc, (const T*)Kuf, Text, Y, sY, (const T*)wT, noise, nrows, ncols, a1, want_grad, Ksc, qbuf);
        hipLaunchKernelGGL((svgp_het_cols_kernel<T>), dim3((unsigned)((SB + 255) / 256)), dim3(256), 0, st, SB, B, M, P, SYc, (const T*)Text, Y, sY, noise,
                           nrows, ncols, (const D*)vard, mat.Kdiag, a1, want_grad, (const T*)qbuf, Eb, dY, dY_shared, dnoise, mat.dKdiag, scal);
        hipLaunchKernelGGL((svgp_het_finalize_kernel<T>), dim3(1), dim3(64), 0, st, S, M, P, (const D*)scal, (const D*)(sc + 0), (const D*)(sc + 1),
                           (const D*)(sc + 2), (const D*)(sc + 3), scaling, a1, logL, dvdir);
        MXF_LAUNCH_CHECK(h);
        if (!want_grad) { hipLaunchKernelGGL(cond_publish_kernel, dim3(1), dim3(1), 0, st, h->cond_dev, cond_slot); return 0; }
        // G' = 1/2 a1 Kuf diag(bs) Kuf^T (-> Psi2 slot), Gw = Kuf (a1 beta.e) (-> R slot), Kuf-side reverse mode from dKuf = Text
        if (het_split) {
            rc = mxf_maxabs_internal(h, M, SB, (const float*)Ksc, SB, hsw + 2, st);
            if (!rc) rc = mxf_split_planes_internal(h, M, SB, (const float*)Ksc, SB, hsS, st, MXF_SPLIT_F16X2, hsw + 2);
            if (!rc) rc = mxf_gemm_split_internal(h, M, M, SB, 1.0, hsS, (int64_t)pl_big, hsK, (int64_t)pl_big, 0.0, (float*)Psi2, M, 0, st, 0, MXF_SPLIT_F16X2, nullptr, 0,
                                                  hsw + 2, hsw + 1);
        } else
        rc = mxf_gemm_internal(h, dtype, 0, 1, M, M, SB, 1.0, Ksc, SB, 0, Kuf, SB, 0, 0.0, Psi2, M, 0, 1, 0, st);
        if (rc) return rc;
        if (SB >= 4096 && M <= 65535) { hipLaunchKernelGGL((rowdot_kernel<T>), dim3((unsigned)M), dim3(256), 0, st, SB, P, (const T*)Kuf, SB, (const T*)Eb, R); rc = 0; }
        else
        rc = mxf_gemm_internal(h, dtype, 0, 0, M, P, SB, 1.0, Kuf, SB, 0, Eb, P, 0, 0.0, R, P, 0, 1, 0, st);
        if (rc) return rc;
        if (use_mat) {
            if (mat.dKuf) MXF_HIP(h, hipMemcpyAsync(mat.dKuf, Text, sizeof(T) * (size_t)M * SB, hipMemcpyDeviceToDevice, st));
        } else {
            rc = mxf_gram_bwd_internal(h, kind, dtype, 1, M, SB, Q, Z, 0, X, 0, ls, ard, 0, var, 0, Text, SB, 0, dZ, dX, dls, dvar, st);
            if (rc) return rc;
        }
    } else if (!want_grad) {
        hipLaunchKernelGGL((svgp_mid_kernel<T>), dim3((unsigned)((SB + 255) / 256)), dim3(256), 0, st, SB, B, M, P, (const T*)Kuf, Text, Y, sY,
                           (const T*)wT, noise, a1, 0, (T*)nullptr, (T*)nullptr, 0, scal);
    } else {
        dnz = sc + 4; dvdir = sc + 5;
        // (dY, dZ, dls, dvar, dX, R were cleared on the second side stream at the start of the call: early_clear)
        // one pass over T: q_n, |e_n|^2, dY, R = Kuf E, and the Kuf-side reverse mode (dX, dZ, dls, dvar) without materialising dKuf
        MXF_T0(h, MXF_T_BWD, st);
        if (fuse_bwd) rc = mxf_svgp_bwd_fuse_finish(h, M, Q, ard, (const float*)ls, (const float*)var, &fza, (float*)dZ, (float*)dls, (float*)dvar, (float*)R, st);
        else
        rc = mxf_svgp_bwd_fused_internal(h, kind, dtype, M, SB, B, Q, P, Z, X, ls, ard, var, Text, het_stream ? (const T*)hys : Y, sY, wT,
                                         het_stream ? (const T*)hnz : noise, a1, dZ, dX, dls, dvar,
                                         dY, dY_shared, R, scal, st, t_blocked,
                                         (use_split && split_mode == MXF_SPLIT_F16X2) ? (const unsigned*)(info2 + 2) : nullptr,
                                         (const unsigned*)(info2 + 3));
        if (rc) return rc;
    }
    if (want_grad && !het) MXF_T1(h, MXF_T_BWD, st);
    MXF_STAGE(h, "reverse pass", st);
    if (het_stream) {
        // sum_n beta_n e_n^2 per sample and the per-row noise gradient: one more pass over the Kfu planes and T''
        if (dnoise) MXF_HIP(h, hipMemsetAsync(dnoise, 0, sizeof(T) * (size_t)B, st));
        hipLaunchKernelGGL(het_stats_kernel, dim3((unsigned)((SB / 16 + 3) / 4)), dim3(256), 0, st, SB, B, M, (const unsigned short*)plKfu, (int64_t)pl_big,
                           (const float*)Text, (const float*)hys, sY, (const float*)noise, (const float*)hrs, (const float*)var, a1, want_grad,
                           (float*)dnoise, hbe2);
        hipLaunchKernelGGL((svgp_finalize_hets_kernel<T>), dim3(1), dim3(64), 0, st, S, B, M, (const D*)scal, (const D*)hbe2, (const D*)hhs, (const D*)noised,
                           (const D*)vard, (const D*)(sc + 0), (const D*)(sc + 1), (const D*)(sc + 2), (const D*)(sc + 3), scaling, a1, logL, dvdir);
        MXF_LAUNCH_CHECK(h);
    } else if (!het) {
        if (whiten) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_join2, 0));      // Phi and the core's Su part (side stream): the value needs tr(C Phi)
        hipLaunchKernelGGL((svgp_finalize_kernel<T>), dim3(1), dim3(64), 0, st, S, B, M, P, (const D*)scal, (const D*)noised, (const D*)vard,
                           (const D*)(sc + 0), (const D*)(sc + 1), (const D*)(sc + 2), (const D*)(sc + 3), scaling, a1, logL, dnz, dvdir,
                           whiten ? (const D*)(sc + 6) : (const D*)nullptr);
        MXF_LAUNCH_CHECK(h);
        if (!want_grad) { hipLaunchKernelGGL(cond_publish_kernel, dim3(1), dim3(1), 0, st, h->cond_dev, cond_slot); return 0; }
    }

    if (het) {
        hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(MM)), dim3(256), 0, st, (int64_t)1, MM, (const T*)Psi2, MM, G, MM);
        hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(MP)), dim3(256), 0, st, (int64_t)1, MP, (const T*)R, MP, Gw, MP);
        // ---- core reverse mode (float64): the Su part on the side stream next to the Kuu part ---------------------------------
        MXF_HIP(h, hipEventRecord(h->ev_fork, st));
        MXF_HIP(h, hipStreamWaitEvent(sd_, h->ev_fork, 0));
        rc = su_reverse(sd_, false);
        if (rc) return rc;
        MXF_HIP(h, hipEventRecord(h->ev_join, sd_));
        rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, G, M, 0, KiSu, M, 0, 0.0, T1, M, 0, 1, 0, st);           // T1 = G Ki Su
        if (rc) return rc;
    } else {
        MXF_HIP(h, hipStreamWaitEvent(st, h->ev_join2, 0));      // side stream: Psi2 -> G, T1, dSu / dW / dSdiag (already done)
        if (side_split) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_k2, 0));      // ... and dKuu0 from the second side stream
        hipLaunchKernelGGL((scale_beta_kernel<T>), dim3(gridn(MP)), dim3(256), 0, st, MP, (const T*)R, (const D*)noised, a1, Gw);
    }
    // main: dKuu = -Ki A_Ki Ki - bP/2 Ki; dmu = Ki Gw - b w
    if (!(early_kuu && !het)) {
    hipLaunchKernelGGL(aki_kernel, dim3(gridn(MM)), dim3(256), 0, st, M, P, (const D*)G, (const D*)T1, (const D*)Gw, (const D*)mud, (const D*)Su, bw, AKi);
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Ki, M, 0, AKi, M, 0, 0.0, T2, M, 0, 1, 0, st);           // T2 = Ki A_Ki
    if (rc) return rc;
    hipLaunchKernelGGL((convert_kernel<D, D>), dim3(gridn(MM)), dim3(256), 0, st, (int64_t)1, MM, (const D*)Ki, MM, dKuu, MM);
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, -1.0, T2, M, 0, Ki, M, 0, -0.5 * bw * P, dKuu, M, 0, 1, 0, st);
    if (rc) return rc;
    }
    if (offpath) hipLaunchKernelGGL((gemv_rows_kernel<D>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, M, P, (const D*)Ki, M, (const D*)Gw, (int64_t)P, dmud, -bw, (const D*)wd);
    else {
    hipLaunchKernelGGL((axpby_kernel<D>), dim3(gridn(MP)), dim3(256), 0, st, MP, -bw, (const D*)wd, 0.0, (const D*)nullptr, dmud);
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, P, M, 1.0, Ki, M, 0, Gw, P, 0, 1.0, dmud, P, 0, 1, 0, st);
    if (rc) return rc;
    }
    if (early_kuu && !het) hipLaunchKernelGGL(dkuu_rank_kernel, dim3(gridn(MM)), dim3(256), 0, st, M, P, (const D*)dmud, (const D*)wd, bw, dKuu);
    // Kuu-side reverse mode in float64, then added to the streaming-side gradients
    FinishArgs fa;
    fa.cnt = 0;
    auto fin = [&](const D* src, const D* src2, void* dst, int64_t n, int acc) {
        fa.src[fa.cnt] = src; fa.src2[fa.cnt] = src2; fa.dst[fa.cnt] = dst; fa.n[fa.cnt] = n; fa.acc[fa.cnt] = acc; ++fa.cnt;
    };
    if (dmu) fin(dmud, nullptr, dmu, MP, 0);
    if (use_mat) {
        if (mat.dKuu) fin(dKuu, nullptr, mat.dKuu, MM, 0);
    } else {
        if (!early_clear) {
            MXF_HIP(h, hipMemsetAsync(dZc, 0, sizeof(D) * M * Q, st));
            MXF_HIP(h, hipMemsetAsync(dlsc, 0, sizeof(D) * lsn, st));
            MXF_HIP(h, hipMemsetAsync(dvc, 0, sizeof(D) * 4, st));
        }
        // (float32 mode: dKuu is symmetric up to the rounding of its float64 products -- the row side of its reverse pass is skipped; float64 keeps both sides)
        rc = mxf_gram_bwd_internal(h, kind, MXF_F64, 1, M, M, Q, Zd, 0, nullptr, 0, lsd, ard, 0, vard, 0, dKuu, M, 0, dZc, nullptr, dlsc, dvc, st,
                                   (early_kuu && !het) ? 1 : 0);
        if (rc) return rc;
        if (dZ) fin(dZc, nullptr, dZ, M * Q, 1);
        if (dls) fin(dlsc, nullptr, dls, lsn, 1);
        if (dvar) fin(dvc, sc + 5, dvar, 1, 1);
    }
    if (dnoise && !het && !het_stream) fin(sc + 4, nullptr, dnoise, 1, 0);
    if (fa.cnt) {
        int64_t nmax = 1;
        for (int i = 0; i < fa.cnt; ++i) nmax = fa.n[i] > nmax ? fa.n[i] : nmax;
        hipLaunchKernelGGL((svgp_finish_kernel<T>), dim3(gridn(nmax), (unsigned)fa.cnt), dim3(256), 0, st, fa);
    }
    MXF_STAGE(h, "core reverse", st);
    hipLaunchKernelGGL(cond_publish_kernel, dim3(1), dim3(1), 0, st, h->cond_dev, cond_slot);
    MXF_HIP(h, hipStreamWaitEvent(st, h->ev_join, 0));     // join the Su chain: every output is ordered on the caller's stream
    MXF_T1(h, MXF_T_CALL, st);
    MXF_STAGE(h, "end", st);
    MXF_STAGE_DUMP(h);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// ================================================================================================ sparse (Titsias) GP
// SparseGPRegressionLogPdf.compute (sparsegp_regression.py:42-108), one sample, in sufficient statistics:
//   C = Kuu + Psi2/s2 (= L A L^T of the reference, so LA = L^-1 chol(C)), a = C^-1 psi1,
//   logL = -P/2 (log|C| - log|Kuu|) - 1/2 (ups/s2 + BP log 2pi + BP log s2) + tr(psi1^T a)/(2 s2^2) - P B var/(2 s2) + P tr(Ki Psi2)/(2 s2)
// reverse mode (verified against autograd of the oracle to 1e-14):
//   GC = -P/2 C^-1 - a a^T/(2 s2^2);  dPsi2 = GC/s2 + P/(2 s2) Ki;  dKuu = GC + P/2 Ki - P/(2 s2) Ki Psi2 Ki;  dpsi1 = a/s2^2
__global__ void sgp_add_psi2_kernel(int64_t n, double* __restrict__ C, const double* __restrict__ Psi2, const double* __restrict__ noise) {
    const double beta = 1.0 / noise[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) C[i] += beta * Psi2[i];
}
template <typename T>
__global__ void sgp_grads_kernel(int64_t M, int P, const double* __restrict__ Ci, const double* __restrict__ Ki, const double* __restrict__ KPK,
                                 const double* __restrict__ a, const double* __restrict__ noise, double gscale, T* __restrict__ G2,
                                 double* __restrict__ GKuu, T* __restrict__ Gpsi1) {
    const double s2 = noise[0], is2 = 1.0 / s2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * M; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / M, j = idx % M;
        double aa = 0;
        for (int p = 0; p < P; ++p) aa += a[i * P + p] * a[j * P + p];
        const double GC = -0.5 * P * Ci[idx] - 0.5 * is2 * is2 * aa;
        G2[idx] = (T)(gscale * 2.0 * (GC * is2 + 0.5 * P * is2 * Ki[idx]));
        GKuu[idx] = gscale * (GC + 0.5 * P * Ki[idx] - 0.5 * P * is2 * KPK[idx]);
        if (idx < M * P) Gpsi1[idx] = (T)(gscale * a[idx] * is2 * is2);
    }
}
// sc: [0] sumlogdiag L  [1] sumlogdiag Lc  [2] <psi1,a>  [3] <Ki,Psi2>  [4] <Ci,Psi2>  [5] a^T Psi2 a  [6] ups
template <typename T>
__global__ void sgp_finalize_kernel(int64_t B, int64_t M, int P, const double* __restrict__ sc, const double* __restrict__ noise,
                                    const double* __restrict__ var, double gscale, T* __restrict__ logL, double* __restrict__ dnoise,
                                    double* __restrict__ dvar_direct, const double* __restrict__ a, T* __restrict__ wv) {
    const double s2 = noise[0], is2 = 1.0 / s2, vk = var[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double l = -(double)P * (sc[1] - sc[0]) - 0.5 * (sc[6] * is2 + (double)B * P * (LOG2PI + log(s2))) + 0.5 * sc[2] * is2 * is2
                         - 0.5 * P * (double)B * vk * is2 + 0.5 * P * sc[3] * is2;
        logL[0] = (T)l;
        const double gcpsi = -0.5 * P * sc[4] - 0.5 * is2 * is2 * sc[5];       // <GC, Psi2>
        if (dnoise) dnoise[0] = gscale * (-gcpsi * is2 * is2 + 0.5 * sc[6] * is2 * is2 - 0.5 * (double)B * P * is2 - sc[2] * is2 * is2 * is2
                                          + 0.5 * P * (double)B * vk * is2 * is2 - 0.5 * P * sc[3] * is2 * is2);
        if (dvar_direct) dvar_direct[0] = gscale * (-0.5 * P * (double)B * is2);
    }
    if (wv) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * P; i += (int64_t)gridDim.x * blockDim.x) wv[i] = (T)(a[i] * is2);
}
// dY = dY (= Kuf^T Gpsi1) - gscale * Y / s2
template <typename T>
__global__ void sgp_dy_kernel(int64_t n, const T* __restrict__ Y, const T* __restrict__ noise, double gscale, T* __restrict__ dY) {
    const T c = (T)gscale / noise[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dY[i] -= c * Y[i];
}

template <typename T>
int sgp_logpdf_typed(mxf_ctx* h, int kind, int dtype, int64_t B, int64_t M, int Q, int P, const T* X, const T* Y, const T* Z, const T* noise,
                     const T* ls, int ard, const T* var, double jitter, double gscale, T* logL, T* wv, T* Lout, T* LAout, int* info,
                     int want_grad, T* dX, T* dY, T* dZ, T* dnoise, T* dls, T* dvar, hipStream_t st) {
    typedef double D;
    const int64_t MM = M * M, MP = M * P;
    const int lsn = ard ? Q : 1;
    size_t need = 0;
    auto acc = [&](size_t n, size_t es) { need += mxf_align(n * es); };
    acc(M * Q, 8); acc(lsn, 8); acc(1, 8); acc(1, 8);
    for (int i = 0; i < 10; ++i) acc(MM, 8);
    acc(MP, 8); acc(MP, 8); acc(MP, 8); acc(16, 8); acc(4, sizeof(int));
    acc((size_t)M * B, sizeof(T)); acc((size_t)M * B, sizeof(T)); acc(MM, sizeof(T)); acc(MP, sizeof(T)); acc(MM, sizeof(T)); acc(MP, sizeof(T));
    acc(M * Q, 8); acc(lsn, 8); acc(4, 8);
    void* ws = mxf_ws(h, need);
    if (!ws) MXF_FAIL(h, -4, "mxf_sgp_logpdf: cannot allocate %zu bytes of scratch", need);
    Carver cv(ws);
    D* Zd = cv.take<D>(M * Q); D* lsd = cv.take<D>(lsn); D* vard = cv.take<D>(1); D* noised = cv.take<D>(1);
    D* Lm = cv.take<D>(MM); D* Linv = cv.take<D>(MM); D* Ki = cv.take<D>(MM); D* Cm = cv.take<D>(MM); D* Lcinv = cv.take<D>(MM);
    D* Ci = cv.take<D>(MM); D* Psi2d = cv.take<D>(MM); D* KPK = cv.take<D>(MM); D* tmp = cv.take<D>(MM); D* GKuu = cv.take<D>(MM);
    D* psi1d = cv.take<D>(MP); D* ad = cv.take<D>(MP); D* PA = cv.take<D>(MP); D* sc = cv.take<D>(16); int* info2 = cv.take<int>(4);
    T* Kuf = cv.take<T>((size_t)M * B); T* Kfu = cv.take<T>((size_t)M * B); T* Psi2 = cv.take<T>(MM); T* psi1 = cv.take<T>(MP);
    T* G2 = cv.take<T>(MM); T* Gpsi1 = cv.take<T>(MP);
    D* dZc = cv.take<D>(M * Q); D* dlsc = cv.take<D>(lsn); D* dvc = cv.take<D>(4);
#define CONV(n, src, dst) hipLaunchKernelGGL((convert_kernel<T, D>), dim3(gridn(n)), dim3(256), 0, st, (int64_t)1, (int64_t)(n), src, (int64_t)(n), dst, (int64_t)(n))
    CONV(M * Q, Z, Zd); CONV(lsn, ls, lsd); CONV(1, var, vard); CONV(1, noise, noised);
    MXF_HIP(h, hipMemsetAsync(sc, 0, 16 * sizeof(D), st));
    // (r04) cond_1(Kuu + jitter I) is published exactly as the SVGP call does (mxf_svgp_cond_slot / mxf_svgp_last_cond): the float32 form of
    // this bound feeds a float32 Psi2 into C = Kuu + Psi2 / s2 and K^-1 Psi2 K^-1 -- ELBO 7e-6 at cond 3e4, a non-PD C at 1e6 -- and the
    // module's guard widens the call to float64 above its limit
    if (!mxf_cond_init(h)) MXF_FAIL(h, -4, "mxf_sgp_logpdf: cannot allocate the condition words");
    MXF_HIP(h, hipMemsetAsync(h->cond_dev, 0, 2 * sizeof(double), st));
    int rc;
    rc = mxf_gram(h, kind, MXF_F64, 1, M, M, Q, Zd, 0, nullptr, 0, lsd, ard, 0, vard, 0, nullptr, 0, jitter, MXF_WRITE, Lm, M, MM, st);   // Kuu :69-72
    if (rc) return rc;
    hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, M, (const double*)Lm, M, h->cond_dev);
    MXF_HIP(h, hipMemcpyAsync(Cm, Lm, MM * sizeof(D), hipMemcpyDeviceToDevice, st));
    rc = mxf_potrf_internal(h, MXF_F64, 1, M, Lm, M, MM, info, st);                                  // L :77
    if (rc) return rc;
    rc = mxf_trtri_internal(h, MXF_F64, 1, M, Lm, M, MM, Linv, M, MM, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, MXF_F64, 1, 0, M, M, M, 1.0, Linv, M, 0, Linv, M, 0, 0.0, Ki, M, 0, 1, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(norm1_sym16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, M, (const double*)Ki, M, h->cond_dev + 1);
    hipLaunchKernelGGL(cond_publish_kernel, dim3(1), dim3(1), 0, st, h->cond_dev, h->cond_host + 2 * h->cond_slot);
    rc = mxf_sumlogdiag_internal(h, MXF_F64, 1, M, Lm, M, MM, sc + 0, st);
    if (rc) return rc;
    // streaming statistics: Psi2 = Kuf Kuf^T (TN on the transposed Gram), psi1 = Kuf Y, ups = |Y|^2
    rc = mxf_gram(h, kind, dtype, 1, M, B, Q, Z, 0, X, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, Kuf, B, 0, st);                 // Kuf :74
    if (rc) return rc;
    rc = mxf_gram(h, kind, dtype, 1, B, M, Q, X, 0, Z, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, Kfu, M, 0, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 1, 0, M, M, B, 1.0, Kfu, M, 0, Kfu, M, 0, 0.0, Psi2, M, 0, 1, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)((M + 31) / 32), (unsigned)((M + 31) / 32), 1), dim3(256), 0, st, Psi2, M, M, MM);
    rc = mxf_gemm_internal(h, dtype, 0, 0, M, P, B, 1.0, Kuf, B, 0, Y, P, 0, 0.0, psi1, P, 0, 1, 0, st);
    if (rc) return rc;
    CONV(MM, (const T*)Psi2, Psi2d); CONV(MP, (const T*)psi1, psi1d);
    hipLaunchKernelGGL((sumsq_kernel<T>), dim3(1), dim3(256), 0, st, B * P, Y, (int64_t)0, sc + 6);
    // C = Kuu + Psi2/s2, Lc = chol(C), Ci, a = Ci psi1
    hipLaunchKernelGGL(sgp_add_psi2_kernel, dim3(gridn(MM)), dim3(256), 0, st, MM, Cm, (const D*)Psi2d, (const D*)noised);
    rc = mxf_potrf_internal(h, MXF_F64, 1, M, Cm, M, MM, info2, st);
    if (rc) return rc;
    rc = mxf_sumlogdiag_internal(h, MXF_F64, 1, M, Cm, M, MM, sc + 1, st);
    if (rc) return rc;
    rc = mxf_trtri_internal(h, MXF_F64, 1, M, Cm, M, MM, Lcinv, M, MM, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, MXF_F64, 1, 0, M, M, M, 1.0, Lcinv, M, 0, Lcinv, M, 0, 0.0, Ci, M, 0, 1, 0, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, P, M, 1.0, Ci, M, 0, psi1d, P, 0, 0.0, ad, P, 0, 1, 0, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, P, M, 1.0, Psi2d, M, 0, ad, P, 0, 0.0, PA, P, 0, 1, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MP)), dim3(256), 0, st, MP, (const D*)psi1d, (const D*)ad, 1.0, sc + 2);
    hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MM)), dim3(256), 0, st, MM, (const D*)Ki, (const D*)Psi2d, 1.0, sc + 3);
    hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MM)), dim3(256), 0, st, MM, (const D*)Ci, (const D*)Psi2d, 1.0, sc + 4);
    hipLaunchKernelGGL((dot_kernel<D>), dim3(dotgrid(MP)), dim3(256), 0, st, MP, (const D*)ad, (const D*)PA, 1.0, sc + 5);
    hipLaunchKernelGGL((sgp_finalize_kernel<T>), dim3(gridn(MP)), dim3(256), 0, st, B, M, P, (const D*)sc, (const D*)noised, (const D*)vard, gscale,
                       logL, want_grad ? sc + 8 : (D*)nullptr, want_grad ? sc + 9 : (D*)nullptr, (const D*)ad, wv);
    // posterior side products (:99-106): L, LA = L^-1 chol(C)
    if (Lout) hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, st, M, M, (const D*)Lm, M, Lout, M);
    if (LAout) {
        rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Linv, M, 0, Cm, M, 0, 0.0, tmp, M, 0, 1, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL((convert_kernel<D, T>), dim3(gridn(MM)), dim3(256), 0, st, M, M, (const D*)tmp, M, LAout, M);
    }
    MXF_LAUNCH_CHECK(h);
    if (!want_grad) return 0;
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, Ki, M, 0, Psi2d, M, 0, 0.0, tmp, M, 0, 1, 0, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, MXF_F64, 0, 0, M, M, M, 1.0, tmp, M, 0, Ki, M, 0, 0.0, KPK, M, 0, 1, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL((sgp_grads_kernel<T>), dim3(gridn(MM)), dim3(256), 0, st, M, P, (const D*)Ci, (const D*)Ki, (const D*)KPK, (const D*)ad,
                       (const D*)noised, gscale, G2, GKuu, Gpsi1);
    // dKuf = 2 dPsi2 Kuf + dpsi1 Y^T  (into the Kfu buffer), then the Gram reverse mode
    T* dKuf = Kfu;
    rc = mxf_gemm_internal(h, dtype, 0, 0, M, B, M, 1.0, G2, M, 0, Kuf, B, 0, 0.0, dKuf, B, 0, 1, 0, st);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 0, 1, M, B, P, 1.0, Gpsi1, P, 0, Y, P, 0, 1.0, dKuf, B, 0, 1, 0, st);
    if (rc) return rc;
    if (dZ) MXF_HIP(h, hipMemsetAsync(dZ, 0, sizeof(T) * M * Q, st));
    if (dls) MXF_HIP(h, hipMemsetAsync(dls, 0, sizeof(T) * lsn, st));
    if (dvar) MXF_HIP(h, hipMemsetAsync(dvar, 0, sizeof(T), st));
    if (dX) MXF_HIP(h, hipMemsetAsync(dX, 0, sizeof(T) * B * Q, st));
    rc = mxf_gram_bwd_internal(h, kind, dtype, 1, M, B, Q, Z, 0, X, 0, ls, ard, 0, var, 0, dKuf, B, 0, dZ, dX, dls, dvar, st);
    if (rc) return rc;
    if (dY) {
        rc = mxf_gemm_internal(h, dtype, 1, 0, B, P, M, 1.0, Kuf, B, 0, Gpsi1, P, 0, 0.0, dY, P, 0, 1, 0, st);     // Kuf^T dpsi1
        if (rc) return rc;
        hipLaunchKernelGGL((sgp_dy_kernel<T>), dim3(gridn(B * P)), dim3(256), 0, st, B * P, Y, noise, gscale, dY);
    }
    MXF_HIP(h, hipMemsetAsync(dZc, 0, sizeof(D) * M * Q, st));
    MXF_HIP(h, hipMemsetAsync(dlsc, 0, sizeof(D) * lsn, st));
    MXF_HIP(h, hipMemsetAsync(dvc, 0, sizeof(D) * 4, st));
    rc = mxf_gram_bwd_internal(h, kind, MXF_F64, 1, M, M, Q, Zd, 0, nullptr, 0, lsd, ard, 0, vard, 0, GKuu, M, 0, dZc, nullptr, dlsc, dvc, st);
    if (rc) return rc;
    if (dZ) hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(gridn(M * Q)), dim3(256), 0, st, M * Q, (T)1, (const D*)dZc, dZ, 1);
    if (dls) hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(gridn(lsn)), dim3(64), 0, st, (int64_t)lsn, (T)1, (const D*)dlsc, dls, 1);
    if (dvar) {
        hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(1), dim3(64), 0, st, (int64_t)1, (T)1, (const D*)dvc, dvar, 1);
        hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(1), dim3(64), 0, st, (int64_t)1, (T)1, (const D*)(sc + 9), dvar, 1);
    }
    if (dnoise) hipLaunchKernelGGL((add_convert_kernel<D, T>), dim3(1), dim3(64), 0, st, (int64_t)1, (T)1, (const D*)(sc + 8), dnoise, 0);
#undef CONV
    MXF_LAUNCH_CHECK(h);
    return 0;
}

}  // namespace

extern "C" int mxf_gp_logpdf(mxf_handle h, int kind, int dtype, int S, int64_t N, int Q, int P,
                             const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                             const void* noise_var, int64_t strideS_noise,
                             const void* lengthscale, int ard, int64_t strideS_ls,
                             const void* variance, int64_t strideS_var, double jitter,
                             void* logL, void* L, void* LinvY, int* info, int want_grad,
                             void* dX, void* dY, void* dnoise, void* dls, void* dvar, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N <= 0 || Q <= 0 || P <= 0) MXF_FAIL(h, -2, "mxf_gp_logpdf: bad shape");
    if (!X || !Y || !noise_var || !lengthscale || !variance || !logL || !L || !LinvY) MXF_FAIL(h, -2, "mxf_gp_logpdf: null argument");
    if (kind > MXF_K_MATERN52) MXF_FAIL(h, -2, "mxf_gp_logpdf: stationary kernels only");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        return gp_logpdf_typed<float>(h, kind, dtype, S, N, Q, P, (const float*)X, strideS_X, (const float*)Y, strideS_Y, (const float*)noise_var,
                                      strideS_noise, (const float*)lengthscale, ard, strideS_ls, (const float*)variance, strideS_var, jitter,
                                      (float*)logL, (float*)L, (float*)LinvY, info, want_grad, (float*)dX, (float*)dY, (float*)dnoise,
                                      (float*)dls, (float*)dvar, st);
    if (dtype == MXF_F64)
        return gp_logpdf_typed<double>(h, kind, dtype, S, N, Q, P, (const double*)X, strideS_X, (const double*)Y, strideS_Y, (const double*)noise_var,
                                       strideS_noise, (const double*)lengthscale, ard, strideS_ls, (const double*)variance, strideS_var, jitter,
                                       (double*)logL, (double*)L, (double*)LinvY, info, want_grad, (double*)dX, (double*)dY, (double*)dnoise,
                                       (double*)dls, (double*)dvar, st);
    MXF_FAIL(h, -2, "mxf_gp_logpdf: bad dtype %d", dtype);
}

static int svgp_dispatch(mxf_handle h, const char* fn, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                         const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y, const void* Z, const void* noise_var,
                         int64_t noise_rows, int noise_cols, const void* qU_mean, const void* qU_cov_W, const void* qU_cov_diag,
                         const void* lengthscale, int ard, const void* variance, double jitter, double scaling, double gscale,
                         void* logL, int* info, int want_grad, void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW,
                         void* dSdiag, void* dls, void* dvar, void* stream) {
    if (!h) return -1;
    if (S <= 0 || B <= 0 || M <= 0 || Q <= 0 || P <= 0) MXF_FAIL(h, -2, "%s: bad shape", fn);
    if (!X || !Y || !Z || !noise_var || !qU_mean || !qU_cov_W || !qU_cov_diag || !lengthscale || !variance || !logL)
        MXF_FAIL(h, -2, "%s: null argument", fn);
    if (kind > MXF_K_MATERN52) MXF_FAIL(h, -2, "%s: stationary kernels only", fn);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        return svgp_logpdf_typed<float>(h, kind, dtype, S, B, M, Q, P, (const float*)X, strideS_X, (const float*)Y, strideS_Y, (const float*)Z,
                                        (const float*)noise_var, noise_rows, noise_cols, (const float*)qU_mean, (const float*)qU_cov_W,
                                        (const float*)qU_cov_diag, (const float*)lengthscale, ard, (const float*)variance, jitter, scaling, gscale,
                                        (float*)logL, info, want_grad, (float*)dX, (float*)dY, (float*)dZ, (float*)dnoise, (float*)dmu, (float*)dW,
                                        (float*)dSdiag, (float*)dls, (float*)dvar, st);
    if (dtype == MXF_F64)
        return svgp_logpdf_typed<double>(h, kind, dtype, S, B, M, Q, P, (const double*)X, strideS_X, (const double*)Y, strideS_Y, (const double*)Z,
                                         (const double*)noise_var, noise_rows, noise_cols, (const double*)qU_mean, (const double*)qU_cov_W,
                                         (const double*)qU_cov_diag, (const double*)lengthscale, ard, (const double*)variance, jitter, scaling,
                                         gscale, (double*)logL, info, want_grad, (double*)dX, (double*)dY, (double*)dZ, (double*)dnoise,
                                         (double*)dmu, (double*)dW, (double*)dSdiag, (double*)dls, (double*)dvar, st);
    MXF_FAIL(h, -2, "%s: bad dtype %d", fn, dtype);
}

extern "C" int mxf_svgp_logpdf(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                               const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                               const void* Z, const void* noise_var, const void* qU_mean, const void* qU_cov_W,
                               const void* qU_cov_diag, const void* lengthscale, int ard, const void* variance,
                               double jitter, double scaling, double gscale,
                               void* logL, int* info, int want_grad,
                               void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag,
                               void* dls, void* dvar, void* stream) {
    return svgp_dispatch(h, "mxf_svgp_logpdf", kind, dtype, S, B, M, Q, P, X, strideS_X, Y, strideS_Y, Z, noise_var, 1, 1, qU_mean, qU_cov_W,
                         qU_cov_diag, lengthscale, ard, variance, jitter, scaling, gscale, logL, info, want_grad, dX, dY, dZ, dnoise, dmu, dW,
                         dSdiag, dls, dvar, stream);
}

// Sampled hyper-parameters / inducing inputs / q(u) (runtime_variable.py:96-118: every operand may carry its own sample axis; the
// reference broadcasts them all to S and evaluates S independent bounds).  ONE call: sample s uses slice s of every operand whose sample
// stride is non-zero (stride 0 = shared), its own (M x M) core included; the samples run back to back on the caller's stream with the
// handle's scratch reused.  Every gradient output carries the sample axis (slice s = gradient of gscale * logL[s] with respect to the
// operands sample s used): a caller whose operand was shared sums its slices.
extern "C" int mxf_svgp_logpdf_sampled(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                                       const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y, const void* Z, int64_t strideS_Z,
                                       const void* noise_var, int64_t strideS_noise, const void* qU_mean, int64_t strideS_mu,
                                       const void* qU_cov_W, int64_t strideS_W, const void* qU_cov_diag, int64_t strideS_sd,
                                       const void* lengthscale, int ard, int64_t strideS_ls, const void* variance, int64_t strideS_var,
                                       double jitter, double scaling, double gscale, void* logL, int* info, int want_grad,
                                       void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag, void* dls, void* dvar,
                                       void* stream) {
    if (!h) return -1;
    if (S <= 0) MXF_FAIL(h, -2, "mxf_svgp_logpdf_sampled: bad shape");
    if (dtype != MXF_F32 && dtype != MXF_F64) MXF_FAIL(h, -2, "mxf_svgp_logpdf_sampled: bad dtype %d", dtype);
    const int64_t e = (int64_t)mxf_esize(dtype), lsn = ard ? Q : 1;
    auto at = [&](const void* p, int64_t off) -> const void* { return p ? (const void*)((const char*)p + off * e) : nullptr; };
    auto atw = [&](void* p, int64_t off) -> void* { return p ? (void*)((char*)p + off * e) : nullptr; };
    for (int s = 0; s < S; ++s) {
        const int rc = svgp_dispatch(h, "mxf_svgp_logpdf_sampled", kind, dtype, 1, B, M, Q, P, at(X, s * strideS_X), 0, at(Y, s * strideS_Y), 0,
                                     at(Z, s * strideS_Z), at(noise_var, s * strideS_noise), 1, 1, at(qU_mean, s * strideS_mu),
                                     at(qU_cov_W, s * strideS_W), at(qU_cov_diag, s * strideS_sd), at(lengthscale, s * strideS_ls), ard,
                                     at(variance, s * strideS_var), jitter, scaling, gscale, atw(logL, s), info ? info + s : nullptr, want_grad,
                                     atw(dX, (int64_t)s * B * Q), atw(dY, (int64_t)s * B * P), atw(dZ, (int64_t)s * M * Q), atw(dnoise, s),
                                     atw(dmu, (int64_t)s * M * P), atw(dW, (int64_t)s * M * M), atw(dSdiag, (int64_t)s * M), atw(dls, (int64_t)s * lsn),
                                     atw(dvar, s), stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int mxf_svgp_logpdf_het(mxf_handle h, int kind, int dtype, int S, int64_t B, int64_t M, int Q, int P,
                                   const void* X, int64_t strideS_X, const void* Y, int64_t strideS_Y,
                                   const void* Z, const void* noise_var, int64_t noise_rows, int noise_cols,
                                   const void* qU_mean, const void* qU_cov_W, const void* qU_cov_diag, const void* lengthscale, int ard,
                                   const void* variance, double jitter, double scaling, double gscale,
                                   void* logL, int* info, int want_grad,
                                   void* dX, void* dY, void* dZ, void* dnoise, void* dmu, void* dW, void* dSdiag,
                                   void* dls, void* dvar, void* stream) {
    return svgp_dispatch(h, "mxf_svgp_logpdf_het", kind, dtype, S, B, M, Q, P, X, strideS_X, Y, strideS_Y, Z, noise_var, noise_rows, noise_cols,
                         qU_mean, qU_cov_W, qU_cov_diag, lengthscale, ard, variance, jitter, scaling, gscale, logL, info, want_grad, dX, dY, dZ,
                         dnoise, dmu, dW, dSdiag, dls, dvar, stream);
}

extern "C" int mxf_svgp_logpdf_mat(mxf_handle h, int dtype, int S, int64_t B, int64_t M, int P, const void* Kuu, const void* Kuf, const void* Kdiag,
                                   const void* Y, int64_t strideS_Y, const void* noise_var, int64_t noise_rows, int noise_cols, const void* qU_mean,
                                   const void* qU_cov_W, const void* qU_cov_diag, double jitter, double scaling, double gscale, void* logL,
                                   int* info, int want_grad, void* dKuu, void* dKuf, void* dKdiag, void* dY, void* dnoise, void* dmu,
                                   void* dW, void* dSdiag, void* stream) {
    if (!h) return -1;
    if (S <= 0 || B <= 0 || M <= 0 || P <= 0) MXF_FAIL(h, -2, "mxf_svgp_logpdf_mat: bad shape");
    if (S > 1 && strideS_Y != B * P) MXF_FAIL(h, -2, "mxf_svgp_logpdf_mat: S > 1 needs contiguous Y samples (S, B, P)");
    const int64_t sYv = S > 1 ? strideS_Y : 0;
    if (!Kuu || !Kuf || !Kdiag || !Y || !noise_var || !qU_mean || !qU_cov_W || !qU_cov_diag || !logL) MXF_FAIL(h, -2, "mxf_svgp_logpdf_mat: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32) {
        SvgpMat<float> m; m.Kuu = (const float*)Kuu; m.Kuf = (const float*)Kuf; m.Kdiag = (const float*)Kdiag;
        m.dKuu = (float*)dKuu; m.dKuf = (float*)dKuf; m.dKdiag = (float*)dKdiag;
        return svgp_logpdf_typed<float>(h, MXF_K_RBF, dtype, S, B, M, 1, P, (const float*)Kuf /*unused X*/, 0, (const float*)Y, sYv, nullptr,
                                        (const float*)noise_var, noise_rows, noise_cols, (const float*)qU_mean, (const float*)qU_cov_W,
                                        (const float*)qU_cov_diag, nullptr, 0, nullptr, jitter, scaling, gscale, (float*)logL, info, want_grad,
                                        nullptr, (float*)dY, nullptr, (float*)dnoise, (float*)dmu, (float*)dW, (float*)dSdiag, nullptr, nullptr, st, m);
    }
    if (dtype == MXF_F64) {
        SvgpMat<double> m; m.Kuu = (const double*)Kuu; m.Kuf = (const double*)Kuf; m.Kdiag = (const double*)Kdiag;
        m.dKuu = (double*)dKuu; m.dKuf = (double*)dKuf; m.dKdiag = (double*)dKdiag;
        return svgp_logpdf_typed<double>(h, MXF_K_RBF, dtype, S, B, M, 1, P, (const double*)Kuf, 0, (const double*)Y, sYv, nullptr,
                                         (const double*)noise_var, noise_rows, noise_cols, (const double*)qU_mean, (const double*)qU_cov_W,
                                         (const double*)qU_cov_diag, nullptr, 0, nullptr, jitter, scaling, gscale, (double*)logL, info, want_grad,
                                         nullptr, (double*)dY, nullptr, (double*)dnoise, (double*)dmu, (double*)dW, (double*)dSdiag, nullptr, nullptr,
                                         st, m);
    }
    MXF_FAIL(h, -2, "mxf_svgp_logpdf_mat: bad dtype %d", dtype);
}

extern "C" int mxf_sgp_logpdf(mxf_handle h, int kind, int dtype, int64_t B, int64_t M, int Q, int P, const void* X, const void* Y,
                              const void* Z, const void* noise_var, const void* lengthscale, int ard, const void* variance,
                              double jitter, double gscale, void* logL, void* wv, void* L, void* LA, int* info, int want_grad,
                              void* dX, void* dY, void* dZ, void* dnoise, void* dls, void* dvar, void* stream) {
    if (!h) return -1;
    if (B <= 0 || M <= 0 || Q <= 0 || P <= 0) MXF_FAIL(h, -2, "mxf_sgp_logpdf: bad shape");
    if (!X || !Y || !Z || !noise_var || !lengthscale || !variance || !logL) MXF_FAIL(h, -2, "mxf_sgp_logpdf: null argument");
    if (kind > MXF_K_MATERN52) MXF_FAIL(h, -2, "mxf_sgp_logpdf: stationary kernels only");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        return sgp_logpdf_typed<float>(h, kind, dtype, B, M, Q, P, (const float*)X, (const float*)Y, (const float*)Z, (const float*)noise_var,
                                       (const float*)lengthscale, ard, (const float*)variance, jitter, gscale, (float*)logL, (float*)wv, (float*)L,
                                       (float*)LA, info, want_grad, (float*)dX, (float*)dY, (float*)dZ, (float*)dnoise, (float*)dls, (float*)dvar, st);
    if (dtype == MXF_F64)
        return sgp_logpdf_typed<double>(h, kind, dtype, B, M, Q, P, (const double*)X, (const double*)Y, (const double*)Z, (const double*)noise_var,
                                        (const double*)lengthscale, ard, (const double*)variance, jitter, gscale, (double*)logL, (double*)wv,
                                        (double*)L, (double*)LA, info, want_grad, (double*)dX, (double*)dY, (double*)dZ, (double*)dnoise,
                                        (double*)dls, (double*)dvar, st);
    MXF_FAIL(h, -2, "mxf_sgp_logpdf: bad dtype %d", dtype);
}

// Blocked Cholesky, triangular solves, triangular inverse and log-det for gfx950, built on the MFMA GEMM.
//
// Replaces MXNet linalg.potrf / trsm / sumlogdiag call sites of the reference
// (gp_regression.py:61-67,172; svgp_regression.py:83-94,151-164; sparsegp_regression.py:77-89).
//
// potrf: two-level blocking.  Outer panels of NBO columns get ONE trailing syrk-style MFMA update
// (K = NBO, lower blocks only) so the trailing matrix is re-read N/NBO times instead of N/64 times;
// inside a panel 64-wide block columns are updated left-looking (MFMA GEMM), their 64x64 diagonal
// block is factored in LDS by one workgroup, and the rows below are solved against it row-per-lane
// (L11 broadcast from LDS, the row held in VGPRs).  Roofline: MFMA bound for the trailing update
// (N^3/3 flops), latency bound on the 64-wide critical path.
#include "common.h"
#include "internal.h"

namespace {

constexpr int NB = 64;     // inner block
constexpr int NBO = 512;   // outer panel

// ---- 64x64 diagonal block Cholesky in LDS, one workgroup per batch item --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void potrf_diag_kernel(T* __restrict__ A, int64_t lda, int64_t sA, int64_t k0, int nb,
                                                          int* __restrict__ info) {
    __shared__ T a[NB][NB + 1];
    __shared__ T col[NB];
    const int tid = threadIdx.x, b = blockIdx.x;
    T* Ab = A + (int64_t)b * sA + k0 * lda + k0;
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, c = e % nb;
        a[i][c] = (c <= i) ? Ab[(int64_t)i * lda + c] : (T)0;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        T d = a[j][j];
        if (!(d > (T)0)) {   // not positive definite (or NaN): record the first failing pivot, keep going finite
            if (tid == 0 && info && info[b] == 0) info[b] = (int)(k0 + j + 1);
            d = (T)1;
        }
        const T rs = (T)1 / sqrt(d);
        if (tid >= j && tid < nb) col[tid] = (tid == j) ? d * rs : a[tid][j] * rs;
        __syncthreads();
        {   // trailing update of the lower triangle: thread (tid&63) owns a column, rows strided by 4
            const int c = j + 1 + (tid & 63);
            if (c < nb) {
                const T cc = col[c];
                for (int i = c + (tid >> 6); i < nb; i += 4) a[i][c] -= col[i] * cc;
            }
        }
        if (tid >= j && tid < nb) a[tid][j] = col[tid];
        __syncthreads();
    }
    for (int e = tid; e < nb * nb; e += 256) {
        const int i = e / nb, c = e % nb;
        Ab[(int64_t)i * lda + c] = (c <= i) ? a[i][c] : (T)0;   // MXNet potrf zeroes the strict upper part
    }
}

// ---- X L11^T = A21 (X overwrites A21): one row per lane, L11 broadcast from LDS ---------------------------
template <typename T>
__global__ __launch_bounds__(128) void solve_rows_kernel(T* __restrict__ A, int64_t lda, int64_t sA, int64_t k0, int nb,
                                                          int64_t r0, int64_t nrows) {
    __shared__ T l[NB][NB + 1];
    __shared__ T t[128][NB + 1];
    const int tid = threadIdx.x, b = blockIdx.y;
    T* Ab = A + (int64_t)b * sA;
    const T* L11 = Ab + k0 * lda + k0;
    const int64_t rb = r0 + (int64_t)blockIdx.x * 128;
    for (int e = tid; e < NB * NB; e += 128) {
        const int i = e / NB, c = e % NB;
        T v = (T)0;
        if (i < nb && c < nb) { if (c <= i) v = L11[(int64_t)i * lda + c]; }
        else if (i == c) v = (T)1;      // identity padding for ragged last block
        l[i][c] = v;
    }
    for (int e = tid; e < 128 * NB; e += 128) {
        const int r = e / NB, c = e % NB;
        t[r][c] = (rb + r < r0 + nrows && c < nb) ? Ab[(rb + r) * lda + k0 + c] : (T)0;
    }
    __syncthreads();
    T x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        T s = t[tid][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = fma(-x[k], l[j][k], s);
        x[j] = s / l[j][j];
        __builtin_amdgcn_sched_barrier(0);   // keep the 2016 broadcast LDS reads from being hoisted (register blow-up)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j) t[tid][j] = x[j];
    __syncthreads();
    for (int e = tid; e < 128 * NB; e += 128) {
        const int r = e / NB, c = e % NB;
        if (rb + r < r0 + nrows && c < nb) Ab[(rb + r) * lda + k0 + c] = t[r][c];
    }
}

// ---- op(L_kk) X = B_k (X overwrites B_k): one right-hand-side column per lane ------------------------------
// TRANS: solve L_kk^T X = B_k by index reversal (P L^T P is lower triangular)
template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void solve_cols_kernel(const T* __restrict__ L, int64_t ldl, int64_t sL, T* __restrict__ B,
                                                          int64_t ldb, int64_t sB, int64_t k0, int nb, int64_t c0, int64_t ncols) {
    __shared__ T l[NB][NB + 1];
    const int tid = threadIdx.x, b = blockIdx.y;
    const T* Lkk = L + (int64_t)b * sL + k0 * ldl + k0;
    T* Bk = B + (int64_t)b * sB + k0 * ldb;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, m = e % NB;
        T v = (T)0;
        if (i < nb && m < nb) {
            if (m <= i) v = TRANS ? Lkk[(int64_t)(nb - 1 - m) * ldl + (nb - 1 - i)] : Lkk[(int64_t)i * ldl + m];
        } else if (i == m) v = (T)1;
        l[i][m] = v;
    }
    __syncthreads();
    const int64_t c = c0 + (int64_t)blockIdx.x * 256 + tid;
    if (c >= c0 + ncols) return;
    // forward substitution in chunks of 8 rows: solved rows are written back to B (in place) and re-read from
    // L1/L2 by later chunks (coalesced across lanes), so only 8 values are live in registers at a time
    constexpr int CH = 8;
    for (int ib = 0; ib < nb; ib += CH) {
        T sacc[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int i = ib + r;
            const int gi = TRANS ? nb - 1 - i : i;
            sacc[r] = (i < nb) ? Bk[(int64_t)gi * ldb + c] : (T)0;
        }
        for (int m = 0; m < ib; ++m) {
            const int gm = TRANS ? nb - 1 - m : m;
            const T xm = Bk[(int64_t)gm * ldb + c];
#pragma unroll
            for (int r = 0; r < CH; ++r) sacc[r] = fma(-xm, l[ib + r][m], sacc[r]);
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
#pragma unroll
            for (int m = 0; m < r; ++m) sacc[r] = fma(-sacc[m], l[ib + r][ib + m], sacc[r]);
            sacc[r] = sacc[r] / l[ib + r][ib + r];
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int i = ib + r;
            const int gi = TRANS ? nb - 1 - i : i;
            if (i < nb) Bk[(int64_t)gi * ldb + c] = sacc[r];
        }
    }
}

template <typename T>
__global__ void zero_upper_kernel(T* A, int64_t n, int64_t lda, int64_t sA) {
    T* a = A + (int64_t)blockIdx.z * sA;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col < n && col > row) a[row * lda + col] = (T)0;
}

template <typename T>
__global__ void set_identity_kernel(T* A, int64_t n, int64_t lda, int64_t sA) {
    T* a = A + (int64_t)blockIdx.z * sA;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col < n) a[row * lda + col] = (col == row) ? (T)1 : (T)0;
}

template <typename T>
__global__ __launch_bounds__(256) void sumlogdiag_kernel(const T* __restrict__ L, int64_t n, int64_t ldl, int64_t sL, T* __restrict__ out) {
    __shared__ T red[16];
    const T* l = L + (int64_t)blockIdx.x * sL;
    T s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += log(fabs(l[i * ldl + i]));
    s = block_sum<T>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

template <typename T>
int potrf_typed(mxf_ctx* h, int dtype, int S, int64_t n, T* A, int64_t lda, int64_t sA, int* info, hipStream_t st) {
    if (info) MXF_HIP(h, hipMemsetAsync(info, 0, sizeof(int) * S, st));
    for (int64_t c0 = 0; c0 < n; c0 += NBO) {
        const int64_t pe = (c0 + NBO < n) ? c0 + NBO : n;   // panel end
        for (int64_t j0 = c0; j0 < pe; j0 += NB) {
            const int nb = (int)((j0 + NB < n) ? NB : n - j0);
            if (j0 > c0) {   // left-looking update of block column j0 with the panel's previous block columns
                int rc = mxf_gemm_internal(h, dtype, 0, 1, n - j0, nb, j0 - c0, -1.0, A + j0 * lda + c0, lda, sA,
                                           A + j0 * lda + c0, lda, sA, 1.0, A + j0 * lda + j0, lda, sA, S, 0, st);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3(S), dim3(256), 0, st, A, lda, sA, j0, nb, info);
            const int64_t below = n - (j0 + nb);
            if (below > 0)
                hipLaunchKernelGGL((solve_rows_kernel<T>), dim3((unsigned)((below + 127) / 128), S), dim3(128), 0, st, A, lda, sA, j0,
                                   nb, j0 + nb, below);
        }
        if (pe < n) {   // trailing update, lower blocks only: A22 -= L21 L21^T with K = panel width
            int rc = mxf_gemm_internal(h, dtype, 0, 1, n - pe, n - pe, pe - c0, -1.0, A + pe * lda + c0, lda, sA,
                                       A + pe * lda + c0, lda, sA, 1.0, A + pe * lda + pe, lda, sA, S, 1, st);
            if (rc) return rc;
        }
    }
    if (n > 1) {
        if (n > 65535) MXF_FAIL(h, -3, "mxf_potrf: n too large");
        hipLaunchKernelGGL((zero_upper_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)n, S), dim3(256), 0, st, A, n, lda, sA);
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// B <- op(L)^-1 B.  rhs_lower: B is (block) lower triangular (used by trtri): block row k only touches columns < (k+1)*NB
template <typename T>
int trsm_typed(mxf_ctx* h, int dtype, int transpose, int S, int64_t n, int64_t nrhs, const T* L, int64_t ldl, int64_t sL,
               T* B, int64_t ldb, int64_t sB, int rhs_lower, hipStream_t st) {
    const int64_t nblk = (n + NB - 1) / NB;
    if (!transpose) {
        for (int64_t kb = 0; kb < nblk; ++kb) {
            const int64_t k0 = kb * NB;
            const int nb = (int)((k0 + NB < n) ? NB : n - k0);
            const int64_t ncols = rhs_lower ? ((k0 + nb < nrhs) ? k0 + nb : nrhs) : nrhs;
            if (k0 > 0) {
                int rc = mxf_gemm_internal(h, dtype, 0, 0, nb, ncols, k0, -1.0, L + k0 * ldl, ldl, sL, B, ldb, sB, 1.0,
                                           B + k0 * ldb, ldb, sB, S, 0, st);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((solve_cols_kernel<T, false>), dim3((unsigned)((ncols + 255) / 256), S), dim3(256), 0, st, L, ldl, sL, B,
                               ldb, sB, k0, nb, (int64_t)0, ncols);
        }
    } else {
        for (int64_t kb = nblk - 1; kb >= 0; --kb) {
            const int64_t k0 = kb * NB;
            const int nb = (int)((k0 + NB < n) ? NB : n - k0);
            const int64_t rem = n - (k0 + nb);
            if (rem > 0) {   // B_k -= L[k+1:, k]^T B[k+1:]
                int rc = mxf_gemm_internal(h, dtype, 1, 0, nb, nrhs, rem, -1.0, L + (k0 + nb) * ldl + k0, ldl, sL,
                                           B + (k0 + nb) * ldb, ldb, sB, 1.0, B + k0 * ldb, ldb, sB, S, 0, st);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((solve_cols_kernel<T, true>), dim3((unsigned)((nrhs + 255) / 256), S), dim3(256), 0, st, L, ldl, sL, B,
                               ldb, sB, k0, nb, (int64_t)0, nrhs);
        }
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

}  // namespace

int mxf_potrf_internal(mxf_ctx* h, int dtype, int S, int64_t n, void* A, int64_t lda, int64_t sA, int* info, hipStream_t st) {
    if (n <= 0 || S <= 0) return 0;
    if (dtype == MXF_F32) return potrf_typed<float>(h, dtype, S, n, (float*)A, lda, sA, info, st);
    if (dtype == MXF_F64) return potrf_typed<double>(h, dtype, S, n, (double*)A, lda, sA, info, st);
    MXF_FAIL(h, -2, "mxf_potrf: bad dtype %d", dtype);
}

int mxf_trsm_internal(mxf_ctx* h, int dtype, int transpose, int S, int64_t n, int64_t nrhs, const void* L, int64_t ldl,
                      int64_t sL, void* B, int64_t ldb, int64_t sB, int rhs_lower, hipStream_t st) {
    if (n <= 0 || nrhs <= 0 || S <= 0) return 0;
    if (dtype == MXF_F32) return trsm_typed<float>(h, dtype, transpose, S, n, nrhs, (const float*)L, ldl, sL, (float*)B, ldb, sB, rhs_lower, st);
    if (dtype == MXF_F64) return trsm_typed<double>(h, dtype, transpose, S, n, nrhs, (const double*)L, ldl, sL, (double*)B, ldb, sB, rhs_lower, st);
    MXF_FAIL(h, -2, "mxf_trsm: bad dtype %d", dtype);
}

int mxf_trtri_internal(mxf_ctx* h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t sL, void* Linv, int64_t ldi,
                       int64_t sI, hipStream_t st) {
    if (n <= 0 || S <= 0) return 0;
    if (n > 65535) MXF_FAIL(h, -3, "mxf_trtri: n too large");
    dim3 g((unsigned)((n + 255) / 256), (unsigned)n, (unsigned)S);
    if (dtype == MXF_F32) hipLaunchKernelGGL((set_identity_kernel<float>), g, dim3(256), 0, st, (float*)Linv, n, ldi, sI);
    else if (dtype == MXF_F64) hipLaunchKernelGGL((set_identity_kernel<double>), g, dim3(256), 0, st, (double*)Linv, n, ldi, sI);
    else MXF_FAIL(h, -2, "mxf_trtri: bad dtype %d", dtype);
    return mxf_trsm_internal(h, dtype, 0, S, n, n, L, ldl, sL, Linv, ldi, sI, 1, st);
}

int mxf_sumlogdiag_internal(mxf_ctx* h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t sL, void* out, hipStream_t st) {
    if (S <= 0) return 0;
    if (dtype == MXF_F32) hipLaunchKernelGGL((sumlogdiag_kernel<float>), dim3(S), dim3(256), 0, st, (const float*)L, n, ldl, sL, (float*)out);
    else if (dtype == MXF_F64) hipLaunchKernelGGL((sumlogdiag_kernel<double>), dim3(S), dim3(256), 0, st, (const double*)L, n, ldl, sL, (double*)out);
    else MXF_FAIL(h, -2, "mxf_sumlogdiag: bad dtype %d", dtype);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int mxf_potrf(mxf_handle h, int dtype, int S, int64_t n, void* A, int64_t lda, int64_t strideS_A, int* info, void* stream) {
    if (!h) return -1;
    if (n < 0 || S < 0 || (n > 0 && (!A || lda < n))) MXF_FAIL(h, -2, "mxf_potrf: bad arguments");
    return mxf_potrf_internal(h, dtype, S, n, A, lda, strideS_A, info, (hipStream_t)stream);
}

extern "C" int mxf_trsm(mxf_handle h, int dtype, int transpose, int S, int64_t n, int64_t nrhs, const void* L, int64_t ldl,
                        int64_t strideS_L, void* B, int64_t ldb, int64_t strideS_B, void* stream) {
    if (!h) return -1;
    if (n < 0 || nrhs < 0 || S < 0 || (n > 0 && nrhs > 0 && (!L || !B || ldl < n || ldb < nrhs))) MXF_FAIL(h, -2, "mxf_trsm: bad arguments");
    return mxf_trsm_internal(h, dtype, transpose, S, n, nrhs, L, ldl, strideS_L, B, ldb, strideS_B, 0, (hipStream_t)stream);
}

extern "C" int mxf_trtri(mxf_handle h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t strideS_L, void* Linv,
                         int64_t ldi, int64_t strideS_I, void* stream) {
    if (!h) return -1;
    if (n < 0 || S < 0 || (n > 0 && (!L || !Linv || ldl < n || ldi < n))) MXF_FAIL(h, -2, "mxf_trtri: bad arguments");
    return mxf_trtri_internal(h, dtype, S, n, L, ldl, strideS_L, Linv, ldi, strideS_I, (hipStream_t)stream);
}

extern "C" int mxf_sumlogdiag(mxf_handle h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t strideS_L, void* out,
                              void* stream) {
    if (!h) return -1;
    if (n < 0 || S < 0 || (n > 0 && (!L || !out))) MXF_FAIL(h, -2, "mxf_sumlogdiag: bad arguments");
    return mxf_sumlogdiag_internal(h, dtype, S, n, L, ldl, strideS_L, out, (hipStream_t)stream);
}

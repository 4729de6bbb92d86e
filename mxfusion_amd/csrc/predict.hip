// Kdiag and the mean / variance predictions as C-ABI composites (mxf_kdiag, mxf_gp_predict, mxf_svgp_predict), for a
// binder that has no PyTorch-side module code.  The same operator sequences as the reference:
//   Kernel.Kdiag                                  kernels/stationary.py:123-124 (variance), linear.py:91-104, static.py:76-86,152-162
//   GPRegressionMeanVariancePrediction.compute    modules/gp_modules/gp_regression.py:146-196
//   SVGPRegressionMeanVariancePrediction.compute  modules/gp_modules/svgp_regression.py:121-189
// built from the library's own kernels (mxf_gram, potrf / trsm, GEMM, coldot).  ONE posterior, S samples of the test inputs: the sample
// axis is folded into the column axis, so every product is one full-width GEMM (the reference broadcasts the posterior to S copies).
#include "common.h"
#include "internal.h"

namespace {

inline unsigned gridn(int64_t n) { int64_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }

// out[s][n] = Kdiag of the kernel at X[s][n]
template <typename T>
__global__ void kdiag_kernel(int kind, int S, int64_t N, int Q, const T* __restrict__ X, int64_t sX, const T* __restrict__ ls, int ard, int64_t sls,
                             const T* __restrict__ var, int64_t svar, T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)S * N; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / N, n = i % N;
        T v;
        if (kind == MXF_K_LINEAR) {           // sum_q variances_q x_q^2 (linear.py:91-104); `ls` carries the variances
            v = 0;
            for (int q = 0; q < Q; ++q) { const T x = X[s * sX + n * Q + q]; v = fma(ls[s * sls + (ard ? q : 0)] * x, x, v); }
        } else {
            v = var[s * svar];                // stationary kernels, Bias, White: the variance
        }
        out[i] = v;
    }
}

// out[i] = base[s(i)] - a[i] (+ b[i]) (+ add[0])      (per-column predictive variance; i over S*N, base per sample or shared)
template <typename T>
__global__ void var_diag_kernel(int64_t n, const T* __restrict__ base, const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ add,
                                T* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T v = base[0] - a[i];
        if (b) v += b[i];
        if (add) v += add[0];
        out[i] = v;
    }
}

template <typename T>
__global__ void add_diag_scalar_kernel(int64_t S, int64_t n, T* __restrict__ A, const T* __restrict__ add, T extra) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S * n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / n, r = i % n;
        A[s * n * n + r * n + r] += (add ? add[0] : (T)0) + extra;
    }
}

template <typename T>
__global__ void diag_embed_kernel(int64_t n, const T* __restrict__ d, T* __restrict__ A) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * n; i += (int64_t)gridDim.x * blockDim.x)
        A[i] = (i / n == i % n) ? d[i / n] : (T)0;
}

template <typename TI, typename TO>
__global__ void pcast_kernel(int64_t n, const TI* __restrict__ src, TO* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (TO)src[i];
}

bool stationary(int kind) { return kind == MXF_K_RBF || kind == MXF_K_MATERN12 || kind == MXF_K_MATERN32 || kind == MXF_K_MATERN52; }

template <typename T>
int gp_predict_typed(mxf_ctx* h, int kind, int dtype, int S, int64_t N, int64_t Nt, int Q, int P, const T* Xc, const T* Xt, const T* ls, int ard,
                     const T* var, const T* L, int64_t ldl, const T* LinvY, const T* noise, int noise_free, int full_cov, T* mean, T* vout,
                     hipStream_t st) {
    const int64_t C = (int64_t)S * Nt;                   // folded columns
    T* V = (T*)mxf_ws(h, sizeof(T) * (size_t)N * C + sizeof(T) * (size_t)C);
    if (!V) MXF_FAIL(h, -4, "mxf_gp_predict: cannot allocate %zu bytes of scratch", sizeof(T) * (size_t)N * C);
    T* cd = V + (size_t)N * C;
    int rc = mxf_gram(h, kind, dtype, 1, N, C, Q, Xc, 0, Xt, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, V, C, 0, st);   // Kxt (N x S Nt) :160
    if (rc) return rc;
    rc = mxf_trsm_internal(h, dtype, 0, 1, N, C, L, ldl, 0, V, C, 0, 0, st);                                                      // V = L^-1 Kxt :161
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 1, 0, C, P, N, 1.0, V, C, 0, LinvY, P, 0, 0.0, mean, P, 0, 1, 0, st);                         // V^T LinvY :162
    if (rc) return rc;
    if (!full_cov) {
        rc = mxf_coldot(h, dtype, 1, N, C, V, C, 0, V, C, 0, cd, st);                                                             // :181
        if (rc) return rc;
        hipLaunchKernelGGL((var_diag_kernel<T>), dim3(gridn(C)), dim3(256), 0, st, C, var, (const T*)cd, (const T*)nullptr, noise_free ? (const T*)nullptr : noise, vout);
    } else {
        // per sample: K(Xt_s, Xt_s) - V_s^T V_s (+ noise I)  :186-190
        rc = mxf_gram(h, kind, dtype, S, Nt, Nt, Q, Xt, Nt * Q, nullptr, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, vout, Nt, Nt * Nt, st);
        if (rc) return rc;
        rc = mxf_gemm_internal(h, dtype, 1, 0, Nt, Nt, N, -1.0, V, C, Nt, V, C, Nt, 1.0, vout, Nt, Nt * Nt, S, 0, st);
        if (rc) return rc;
        if (!noise_free) hipLaunchKernelGGL((add_diag_scalar_kernel<T>), dim3(gridn(C)), dim3(256), 0, st, (int64_t)S, Nt, vout, noise, (T)0);
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

template <typename T>
int svgp_predict_typed(mxf_ctx* h, int kind, int dtype, int S, int64_t M, int64_t Nt, int Q, int P, const T* Z, const T* Xt, const T* ls, int ard,
                       const T* var, const T* mu, const T* W, const T* sdiag, const T* noise, double jitter, int noise_free, int full_cov, T* mean,
                       T* vout, int* info, hipStream_t st, size_t ws_off = 0) {
    const int64_t C = (int64_t)S * Nt, MM = M * M;
    const size_t need = sizeof(T) * ((size_t)4 * MM + 2 * (size_t)M * P + 2 * (size_t)M * C + 2 * (size_t)C) + 64;
    char* ws0 = (char*)mxf_ws(h, ws_off + need);
    if (!ws0) MXF_FAIL(h, -4, "mxf_svgp_predict: cannot allocate %zu bytes of scratch", ws_off + need);
    T* base = (T*)(ws0 + ws_off);
    T* Lm = base; T* Su = Lm + MM; T* LinvLs = Su + MM; T* LSL = LinvLs + MM;
    T* Linvmu = LSL + MM; T* wv = Linvmu + M * P;
    T* V = wv + M * P; T* tmp = V + (size_t)M * C; T* cd1 = tmp + (size_t)M * C; T* cd2 = cd1 + C;
    // everything that does not depend on the test inputs (:141-155)
    hipLaunchKernelGGL((diag_embed_kernel<T>), dim3(gridn(MM)), dim3(256), 0, st, M, sdiag, Su);
    int rc = mxf_gemm_internal(h, dtype, 0, 1, M, M, M, 1.0, W, M, 0, W, M, 0, 1.0, Su, M, 0, 1, 0, st);                             // S = W W^T + diag :145
    if (rc) return rc;
    rc = mxf_gram(h, kind, dtype, 1, M, M, Q, Z, 0, nullptr, 0, ls, ard, 0, var, 0, nullptr, 0, jitter, MXF_WRITE, Lm, M, 0, st);     // Kuu (+ jitter) :146-148
    if (rc) return rc;
    rc = mxf_potrf_internal(h, dtype, 1, M, Lm, M, 0, info, st);                                                                     // L :149
    if (rc) return rc;
    rc = mxf_potrf_internal(h, dtype, 1, M, Su, M, 0, info ? info + 1 : nullptr, st);                                                // Ls :150
    if (rc) return rc;
    MXF_HIP(h, hipMemcpyAsync(LinvLs, Su, sizeof(T) * MM, hipMemcpyDeviceToDevice, st));
    rc = mxf_trsm_internal(h, dtype, 0, 1, M, M, Lm, M, 0, LinvLs, M, 0, 0, st);                                                     // L^-1 Ls :151
    if (rc) return rc;
    MXF_HIP(h, hipMemcpyAsync(Linvmu, mu, sizeof(T) * M * P, hipMemcpyDeviceToDevice, st));
    rc = mxf_trsm_internal(h, dtype, 0, 1, M, P, Lm, M, 0, Linvmu, P, 0, 0, st);                                                     // L^-1 mu :152
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 0, 1, M, M, M, 1.0, LinvLs, M, 0, LinvLs, M, 0, 0.0, LSL, M, 0, 1, 0, st);                      // (L^-1 Ls)(L^-1 Ls)^T :153
    if (rc) return rc;
    MXF_HIP(h, hipMemcpyAsync(wv, Linvmu, sizeof(T) * M * P, hipMemcpyDeviceToDevice, st));
    rc = mxf_trsm_internal(h, dtype, 1, 1, M, P, Lm, M, 0, wv, P, 0, 0, st);                                                         // L^-T L^-1 mu :154
    if (rc) return rc;
    // test inputs
    rc = mxf_gram(h, kind, dtype, 1, M, C, Q, Z, 0, Xt, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, V, C, 0, st);             // Kxt :157
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 1, 0, C, P, M, 1.0, V, C, 0, wv, P, 0, 0.0, mean, P, 0, 1, 0, st);                               // Kxt^T wv :158
    if (rc) return rc;
    rc = mxf_trsm_internal(h, dtype, 0, 1, M, C, Lm, M, 0, V, C, 0, 0, st);                                                          // V = L^-1 Kxt :162
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 0, 0, M, C, M, 1.0, LSL, M, 0, V, C, 0, 0.0, tmp, C, 0, 1, 0, st);                               // :165
    if (rc) return rc;
    if (!full_cov) {
        rc = mxf_coldot(h, dtype, 1, M, C, V, C, 0, V, C, 0, cd1, st);
        if (rc) return rc;
        rc = mxf_coldot(h, dtype, 1, M, C, tmp, C, 0, V, C, 0, cd2, st);
        if (rc) return rc;
        hipLaunchKernelGGL((var_diag_kernel<T>), dim3(gridn(C)), dim3(256), 0, st, C, var, (const T*)cd1, (const T*)cd2, noise_free ? (const T*)nullptr : noise, vout);   // :166-172
    } else {
        rc = mxf_gram(h, kind, dtype, S, Nt, Nt, Q, Xt, Nt * Q, nullptr, 0, ls, ard, 0, var, 0, nullptr, 0, 0.0, MXF_WRITE, vout, Nt, Nt * Nt, st);
        if (rc) return rc;
        rc = mxf_gemm_internal(h, dtype, 1, 0, Nt, Nt, M, -1.0, V, C, Nt, V, C, Nt, 1.0, vout, Nt, Nt * Nt, S, 0, st);               // :176-178
        if (rc) return rc;
        rc = mxf_gemm_internal(h, dtype, 1, 0, Nt, Nt, M, 1.0, V, C, Nt, tmp, C, Nt, 1.0, vout, Nt, Nt * Nt, S, 0, st);
        if (rc) return rc;
        if (!noise_free) hipLaunchKernelGGL((add_diag_scalar_kernel<T>), dim3(gridn(C)), dim3(256), 0, st, (int64_t)S, Nt, vout, noise, (T)0);
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

}  // namespace

extern "C" int mxf_kdiag(mxf_handle h, int kind, int dtype, int S, int64_t N, int Q, const void* X, int64_t strideS_X, const void* lengthscale, int ard,
                         int64_t strideS_ls, const void* variance, int64_t strideS_var, void* out, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N <= 0) return 0;
    if (!out || (kind == MXF_K_LINEAR ? (!X || !lengthscale) : !variance)) MXF_FAIL(h, -2, "mxf_kdiag: null argument");
    if (kind < MXF_K_RBF || kind > MXF_K_WHITE) MXF_FAIL(h, -2, "mxf_kdiag: unknown kernel kind %d", kind);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        hipLaunchKernelGGL((kdiag_kernel<float>), dim3(gridn((int64_t)S * N)), dim3(256), 0, st, kind, S, N, Q, (const float*)X, strideS_X, (const float*)lengthscale, ard,
                           strideS_ls, (const float*)variance, strideS_var, (float*)out);
    else if (dtype == MXF_F64)
        hipLaunchKernelGGL((kdiag_kernel<double>), dim3(gridn((int64_t)S * N)), dim3(256), 0, st, kind, S, N, Q, (const double*)X, strideS_X, (const double*)lengthscale,
                           ard, strideS_ls, (const double*)variance, strideS_var, (double*)out);
    else MXF_FAIL(h, -2, "mxf_kdiag: bad dtype %d", dtype);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int mxf_gp_predict(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t Nt, int Q, int P, const void* X_cond, const void* X_test,
                              const void* lengthscale, int ard, const void* variance, const void* L, int64_t ldl, const void* LinvY,
                              const void* noise_var, int noise_free, int full_cov, void* mean_out, void* var_out, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N <= 0 || Nt <= 0 || Q <= 0 || P <= 0) MXF_FAIL(h, -2, "mxf_gp_predict: bad shape");
    if (!stationary(kind)) MXF_FAIL(h, -2, "mxf_gp_predict: stationary kernels only (kind %d); compose the other kinds from mxf_gram / mxf_trsm / mxf_gemm", kind);
    if (!X_cond || !X_test || !lengthscale || !variance || !L || !LinvY || !mean_out || !var_out || (!noise_free && !noise_var))
        MXF_FAIL(h, -2, "mxf_gp_predict: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        return gp_predict_typed<float>(h, kind, dtype, S, N, Nt, Q, P, (const float*)X_cond, (const float*)X_test, (const float*)lengthscale, ard, (const float*)variance,
                                       (const float*)L, ldl, (const float*)LinvY, (const float*)noise_var, noise_free, full_cov, (float*)mean_out, (float*)var_out, st);
    if (dtype == MXF_F64)
        return gp_predict_typed<double>(h, kind, dtype, S, N, Nt, Q, P, (const double*)X_cond, (const double*)X_test, (const double*)lengthscale, ard,
                                        (const double*)variance, (const double*)L, ldl, (const double*)LinvY, (const double*)noise_var, noise_free, full_cov,
                                        (double*)mean_out, (double*)var_out, st);
    MXF_FAIL(h, -2, "mxf_gp_predict: bad dtype %d", dtype);
}

extern "C" int mxf_svgp_predict(mxf_handle h, int kind, int dtype, int S, int64_t M, int64_t Nt, int Q, int P, const void* Z, const void* X_test,
                                const void* lengthscale, int ard, const void* variance, const void* qU_mean, const void* qU_cov_W,
                                const void* qU_cov_diag, const void* noise_var, double jitter, int noise_free, int full_cov, void* mean_out,
                                void* var_out, void* info, void* stream) {
    if (!h) return -1;
    if (S <= 0 || M <= 0 || Nt <= 0 || Q <= 0 || P <= 0) MXF_FAIL(h, -2, "mxf_svgp_predict: bad shape");
    if (!stationary(kind)) MXF_FAIL(h, -2, "mxf_svgp_predict: stationary kernels only (kind %d)", kind);
    if (!Z || !X_test || !lengthscale || !variance || !qU_mean || !qU_cov_W || !qU_cov_diag || !mean_out || !var_out || (!noise_free && !noise_var))
        MXF_FAIL(h, -2, "mxf_svgp_predict: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (info) MXF_HIP(h, hipMemsetAsync(info, 0, 2 * sizeof(int), st));
    if (dtype == MXF_F32) {
        // r04: float32 predictions are EVALUATED in float64 (inputs widened, results narrowed).  The reference factors Kuu in the model's
        // dtype (svgp_regression.py:146-154); in float32 that loses cond(Kuu) 2^-24 of the posterior moments -- 1e-3 .. 1e-1 at the condition
        // numbers a trained model has (2e4 .. 1e6), where north_star asks for 1e-5.  Prediction is not the hot path (one M^3 core + 2 M^2 Nt
        // flops per call); the training call has kept its (M x M) core in float64 since r01 for the same reason.
        const int64_t C = (int64_t)S * Nt, MM = M * M;
        const int lsn = ard ? Q : 1;
        const size_t nvo = full_cov ? (size_t)S * Nt * Nt : (size_t)C;
        size_t off = 0;
        auto take = [&](size_t n) { const size_t o = off; off += mxf_align(n * sizeof(double)); return o; };
        const size_t oZ = take(M * Q), oX = take((size_t)C * Q), ols = take(lsn), ovr = take(1), omu = take(M * P), oW = take(MM), osd = take(M), onz = take(1),
                     omean = take((size_t)C * P), ovout = take(nvo);
        const size_t need_d = sizeof(double) * ((size_t)4 * MM + 2 * (size_t)M * P + 2 * (size_t)M * C + 2 * (size_t)C) + 64;
        char* ws = (char*)mxf_ws(h, off + need_d);
        if (!ws) MXF_FAIL(h, -4, "mxf_svgp_predict: cannot allocate %zu bytes of scratch", off + need_d);
        auto up = [&](const void* src, size_t o, size_t n) {
            if (src && n) hipLaunchKernelGGL((pcast_kernel<float, double>), dim3(gridn((int64_t)n)), dim3(256), 0, st, (int64_t)n, (const float*)src, (double*)(ws + o));
        };
        up(Z, oZ, M * Q); up(X_test, oX, (size_t)C * Q); up(lengthscale, ols, lsn); up(variance, ovr, 1); up(qU_mean, omu, M * P); up(qU_cov_W, oW, MM);
        up(qU_cov_diag, osd, M); up(noise_var, onz, 1);
        const int rc = svgp_predict_typed<double>(h, kind, MXF_F64, S, M, Nt, Q, P, (const double*)(ws + oZ), (const double*)(ws + oX), (const double*)(ws + ols), ard,
                                                  (const double*)(ws + ovr), (const double*)(ws + omu), (const double*)(ws + oW), (const double*)(ws + osd),
                                                  noise_var ? (const double*)(ws + onz) : nullptr, jitter, noise_free, full_cov, (double*)(ws + omean),
                                                  (double*)(ws + ovout), (int*)info, st, off);
        if (rc) return rc;
        hipLaunchKernelGGL((pcast_kernel<double, float>), dim3(gridn((int64_t)C * P)), dim3(256), 0, st, (int64_t)C * P, (const double*)(ws + omean), (float*)mean_out);
        hipLaunchKernelGGL((pcast_kernel<double, float>), dim3(gridn((int64_t)nvo)), dim3(256), 0, st, (int64_t)nvo, (const double*)(ws + ovout), (float*)var_out);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    if (dtype == MXF_F64)
        return svgp_predict_typed<double>(h, kind, dtype, S, M, Nt, Q, P, (const double*)Z, (const double*)X_test, (const double*)lengthscale, ard,
                                          (const double*)variance, (const double*)qU_mean, (const double*)qU_cov_W, (const double*)qU_cov_diag,
                                          (const double*)noise_var, jitter, noise_free, full_cov, (double*)mean_out, (double*)var_out, (int*)info, st);
    MXF_FAIL(h, -2, "mxf_svgp_predict: bad dtype %d", dtype);
}

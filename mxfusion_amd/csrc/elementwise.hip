// Elementwise / reduction kernels of the Monte-Carlo ELBO loop (gfx950): softplus transform,
// Normal log-pdf with fused reverse mode and wavefront-shuffle reductions, reparameterised sampling
// and its reverse mode, MXNet-Adam.  All HBM-bound streaming kernels: 16-byte loads where the layout
// allows, grid-stride, one atomic per block for the scalar sums.
//
// Replaces: PositiveTransformation (components/variables/var_trans.py:63-91), Normal.log_pdf_impl /
// draw_samples_impl (components/distributions/normal.py:52-92) + the sum(mean_S(.)) of
// models/factor_graph.py:223, and gluon.Trainer.step(adam) of inference/batch_loop.py:46-60.
#include "common.h"

namespace {

constexpr double LOG2PI = 1.8378770664093453;

template <typename T> __device__ __forceinline__ T softplus_f(T x) {
    // log(1+exp(x)) = max(x,0) + log1p(exp(-|x|))  (overflow-safe form of MXNet's softrelu)
    return fmax(x, (T)0) + log1p(exp(-fabs(x)));
}
template <typename T> __device__ __forceinline__ T sigmoid_f(T x) {
    return x >= (T)0 ? (T)1 / ((T)1 + exp(-x)) : exp(x) / ((T)1 + exp(x));
}

template <typename T>
__global__ void softplus_fwd_kernel(int64_t n, const T* __restrict__ x, T* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = softplus_f<T>(x[i]);
}
template <typename T>
__global__ void softplus_bwd_kernel(int64_t n, const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] += dy[i] * sigmoid_f<T>(x[i]);
}

template <typename T>
__global__ void reparam_kernel(int S, int64_t n, const T* __restrict__ mean, const T* __restrict__ var, const T* __restrict__ eps,
                               T* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T m = mean[i], sd = sqrt(var[i]);
        for (int s = 0; s < S; ++s) x[(int64_t)s * n + i] = fma(eps[(int64_t)s * n + i], sd, m);
    }
}

template <typename T>
__global__ void reparam_bwd_kernel(int S, int64_t n, const T* __restrict__ var, const T* __restrict__ eps, const T* __restrict__ dx,
                                   T* __restrict__ dmean, T* __restrict__ dvar) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T gm = 0, gv = 0;
        for (int s = 0; s < S; ++s) {
            const T g = dx[(int64_t)s * n + i];
            gm += g;
            gv = fma(g, eps[(int64_t)s * n + i], gv);
        }
        if (dmean) dmean[i] += gm;
        if (dvar) dvar[i] += gv * (T)0.5 / sqrt(var[i]);
    }
}

// out += scale * sum logN(x|mean,var); dx += scale * dlogN/dx ...; thread i owns element i for all S samples,
// so dmean/dvar need no atomics when they are per-element; broadcast (single-element) mean/var are block-reduced.
template <typename T>
__global__ __launch_bounds__(256) void normal_logpdf_kernel(int S, int64_t n, const T* __restrict__ x, const T* __restrict__ mean,
                                                            int64_t n_mean, const T* __restrict__ var, int64_t n_var, T scale,
                                                            T* __restrict__ out, T* __restrict__ dx, T* __restrict__ dmean,
                                                            T* __restrict__ dvar) {
    __shared__ T red[16];
    T acc = 0, gm_b = 0, gv_b = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T m = mean[n_mean == 1 ? 0 : i], v = var[n_var == 1 ? 0 : i];
        const T iv = (T)1 / v;
        const T c = (T)(-0.5 * LOG2PI) - (T)0.5 * log(v);
        T gm = 0, gv = 0;
        for (int s = 0; s < S; ++s) {
            const T d = x[(int64_t)s * n + i] - m;
            acc += c - (T)0.5 * d * d * iv;
            const T g = -d * iv * scale;                 // d/dx
            if (dx) dx[(int64_t)s * n + i] += g;
            gm -= g;                                      // d/dmean = -d/dx
            gv += scale * (T)0.5 * (d * d * iv * iv - iv);   // d/dvar
        }
        if (dmean) { if (n_mean == 1) gm_b += gm; else dmean[i] += gm; }
        if (dvar) { if (n_var == 1) gv_b += gv; else dvar[i] += gv; }
    }
    acc = block_sum<T>(acc, red);
    if (threadIdx.x == 0 && out) atomic_add(out, acc * scale);
    if (dmean && n_mean == 1) { gm_b = block_sum<T>(gm_b, red); if (threadIdx.x == 0) atomic_add(dmean, gm_b); }
    if (dvar && n_var == 1) { gv_b = block_sum<T>(gv_b, red); if (threadIdx.x == 0) atomic_add(dvar, gv_b); }
}

template <typename T>
__global__ void adam_kernel(int64_t n, T* __restrict__ w, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v, T lr_t, T b1,
                            T b2, T eps, T rescale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T gi = g[i] * rescale;
        const T mi = b1 * m[i] + ((T)1 - b1) * gi;
        const T vi = b2 * v[i] + ((T)1 - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] -= lr_t * mi / (sqrt(vi) + eps);
    }
}

// out[s][n] = sum_m A[s][m][n] * B[s][m][n]   (F.sum(A*B, axis=-2): gp_regression.py:181, svgp_regression.py:166-169)
template <typename T>
__global__ void coldot_kernel(int64_t M, int64_t N, const T* __restrict__ A, int64_t lda, int64_t sA, const T* __restrict__ B, int64_t ldb,
                              int64_t sB, T* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const T* a = A + (int64_t)blockIdx.y * sA;
    const T* b = B + (int64_t)blockIdx.y * sB;
    T acc = 0;
    for (int64_t m = 0; m < M; ++m) acc = fma(a[m * lda + n], b[m * ldb + n], acc);
    out[(int64_t)blockIdx.y * N + n] = acc;
}

// few columns (the rollout's one test point per trajectory: N = 64, M = 1000): a thread per column leaves the chip idle and walks the rows
// serially (224 us at that shape); here a workgroup owns 16 columns and spreads the rows over 16 lanes each, combined through LDS
template <typename T>
__global__ __launch_bounds__(256) void coldot_small_kernel(int64_t M, int64_t N, const T* __restrict__ A, int64_t lda, int64_t sA,
                                                           const T* __restrict__ B, int64_t ldb, int64_t sB, T* __restrict__ out) {
    __shared__ T part[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int64_t n = (int64_t)blockIdx.x * 16 + cx;
    const T* a = A + (int64_t)blockIdx.y * sA;
    const T* b = B + (int64_t)blockIdx.y * sB;
    T acc = 0;
    if (n < N)
        for (int64_t m = ry; m < M; m += 16) acc = fma(a[m * lda + n], b[m * ldb + n], acc);
    part[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && n < N) {
        T t = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][cx];
        out[(int64_t)blockIdx.y * N + n] = t;
    }
}

// make_diagonal (util/customop.py:22-81): out[b][i][j] = (i == j) ? a[b][i] : 0; its reverse mode extracts the diagonal of the cotangent
template <typename T>
__global__ void make_diagonal_kernel(int64_t total, int64_t n, const T* __restrict__ a, T* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e % n, i = (e / n) % n, b = e / (n * n);
        out[e] = (i == j) ? a[b * n + i] : (T)0;
    }
}
template <typename T>
__global__ void diag_of_kernel(int64_t total, int64_t n, const T* __restrict__ g, T* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % n, b = e / n;
        out[e] = g[(b * n + i) * n + i];
    }
}

// kernels that end in ONE same-address atomic per workgroup (the scalar sums of normal_logpdf_kernel): 2048 of them serialise at the L2
// (~15 ns apiece: 33 us for the 2 M-element log-pdf of a 4-sample step, of which the data take 6) -- two workgroups per CU
inline unsigned grid_for_reduce(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return (unsigned)b;
}
inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

}  // namespace

#define DISPATCH(h, dtype, name, CALLF, CALLD)             \
    do {                                                   \
        if (dtype == MXF_F32) { CALLF; }                   \
        else if (dtype == MXF_F64) { CALLD; }              \
        else MXF_FAIL(h, -2, name ": bad dtype %d", dtype); \
        MXF_LAUNCH_CHECK(h);                               \
        return 0;                                          \
    } while (0)

extern "C" int mxf_softplus_fwd(mxf_handle h, int dtype, int64_t n, const void* x, void* y, void* stream) {
    if (!h) return -1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_softplus_fwd",
             hipLaunchKernelGGL((softplus_fwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, n, (const float*)x, (float*)y),
             hipLaunchKernelGGL((softplus_fwd_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, n, (const double*)x, (double*)y));
}

extern "C" int mxf_softplus_bwd(mxf_handle h, int dtype, int64_t n, const void* x, const void* dy, void* dx_acc, void* stream) {
    if (!h) return -1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_softplus_bwd",
             hipLaunchKernelGGL((softplus_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, n, (const float*)x, (const float*)dy, (float*)dx_acc),
             hipLaunchKernelGGL((softplus_bwd_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, n, (const double*)x, (const double*)dy, (double*)dx_acc));
}

extern "C" int mxf_normal_reparam(mxf_handle h, int dtype, int S, int64_t n, const void* mean, const void* var, const void* eps,
                                  void* x, void* stream) {
    if (!h) return -1;
    if (n <= 0 || S <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_normal_reparam",
             hipLaunchKernelGGL((reparam_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, S, n, (const float*)mean, (const float*)var, (const float*)eps, (float*)x),
             hipLaunchKernelGGL((reparam_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, S, n, (const double*)mean, (const double*)var, (const double*)eps, (double*)x));
}

extern "C" int mxf_normal_reparam_bwd(mxf_handle h, int dtype, int S, int64_t n, const void* var, const void* eps, const void* dx,
                                      void* dmean_acc, void* dvar_acc, void* stream) {
    if (!h) return -1;
    if (n <= 0 || S <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_normal_reparam_bwd",
             hipLaunchKernelGGL((reparam_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, S, n, (const float*)var, (const float*)eps, (const float*)dx, (float*)dmean_acc, (float*)dvar_acc),
             hipLaunchKernelGGL((reparam_bwd_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, S, n, (const double*)var, (const double*)eps, (const double*)dx, (double*)dmean_acc, (double*)dvar_acc));
}

extern "C" int mxf_normal_logpdf(mxf_handle h, int dtype, int S, int64_t n, const void* x, const void* mean, int64_t n_mean,
                                 const void* var, int64_t n_var, double scale, void* out_acc, void* dx_acc, void* dmean_acc,
                                 void* dvar_acc, void* stream) {
    if (!h) return -1;
    if (n <= 0 || S <= 0) return 0;
    if ((n_mean != 1 && n_mean != n) || (n_var != 1 && n_var != n)) MXF_FAIL(h, -2, "mxf_normal_logpdf: mean/var must have 1 or n elements");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_normal_logpdf",
             hipLaunchKernelGGL((normal_logpdf_kernel<float>), dim3(grid_for_reduce(n)), dim3(256), 0, st, S, n, (const float*)x, (const float*)mean, n_mean, (const float*)var, n_var, (float)scale, (float*)out_acc, (float*)dx_acc, (float*)dmean_acc, (float*)dvar_acc),
             hipLaunchKernelGGL((normal_logpdf_kernel<double>), dim3(grid_for_reduce(n)), dim3(256), 0, st, S, n, (const double*)x, (const double*)mean, n_mean, (const double*)var, n_var, scale, (double*)out_acc, (double*)dx_acc, (double*)dmean_acc, (double*)dvar_acc));
}

extern "C" int mxf_adam_step(mxf_handle h, int dtype, int64_t n, void* w, const void* g, void* m, void* v, double lr, double beta1,
                             double beta2, double epsilon, double rescale_grad, int t, void* stream) {
    if (!h) return -1;
    if (n <= 0) return 0;
    if (t < 1) MXF_FAIL(h, -2, "mxf_adam_step: t must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const double lr_t = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    DISPATCH(h, dtype, "mxf_adam_step",
             hipLaunchKernelGGL((adam_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, n, (float*)w, (const float*)g, (float*)m, (float*)v, (float)lr_t, (float)beta1, (float)beta2, (float)epsilon, (float)rescale_grad),
             hipLaunchKernelGGL((adam_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, n, (double*)w, (const double*)g, (double*)m, (double*)v, lr_t, beta1, beta2, epsilon, rescale_grad));
}

// out[0] = sum_i g[i] if all g[i] agree to 1e-6 relative, NaN otherwise (n small: the upstream gradient of the per-sample log-pdfs)
template <typename T>
__global__ __launch_bounds__(64) void uniform_sum_kernel(int64_t n, const T* __restrict__ g, T* __restrict__ out) {
    T s = 0, lo = g[0], hi = g[0];
    for (int64_t i = threadIdx.x; i < n; i += 64) { const T v = g[i]; s += v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const T l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    if (threadIdx.x == 0) {
        const T mean = s / (T)n, a = mean < 0 ? -mean : mean;
        out[0] = (hi - lo <= (T)1e-6 * a) ? s : (T)NAN;
    }
}

extern "C" int mxf_uniform_sum(mxf_handle h, int dtype, int64_t n, const void* g, void* out, void* stream) {
    if (!h) return -1;
    if (n <= 0 || !g || !out) MXF_FAIL(h, -2, "mxf_uniform_sum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_uniform_sum",
             hipLaunchKernelGGL((uniform_sum_kernel<float>), dim3(1), dim3(64), 0, st, n, (const float*)g, (float*)out),
             hipLaunchKernelGGL((uniform_sum_kernel<double>), dim3(1), dim3(64), 0, st, n, (const double*)g, (double*)out));
}

// MXNet SGD (optimizer 'sgd' of gluon.Trainer): g = rescale * grad + wd * w; momentum == 0: w -= lr g; else mom = momentum mom - lr g, w += mom
template <typename T>
__global__ void sgd_kernel(int64_t n, T* __restrict__ w, const T* __restrict__ g, T* __restrict__ mom, T lr, T momentum, T wd, T rescale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T gi = g[i] * rescale + wd * w[i];
        if (mom) { const T mi = momentum * mom[i] - lr * gi; mom[i] = mi; w[i] += mi; }
        else w[i] -= lr * gi;
    }
}

extern "C" int mxf_sgd_step(mxf_handle h, int dtype, int64_t n, void* w, const void* g, void* mom, double lr, double momentum, double wd,
                            double rescale_grad, void* stream) {
    if (!h) return -1;
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH(h, dtype, "mxf_sgd_step",
             hipLaunchKernelGGL((sgd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, n, (float*)w, (const float*)g, (float*)mom, (float)lr, (float)momentum, (float)wd, (float)rescale_grad),
             hipLaunchKernelGGL((sgd_kernel<double>), dim3(grid_for(n)), dim3(256), 0, st, n, (double*)w, (const double*)g, (double*)mom, lr, momentum, wd, rescale_grad));
}

// The other optimisers gluon.Trainer is commonly driven with by name (batch_loop.py:46-49 hands `optimizer` to the Trainer): MXNet's
// update rules as documented for its 1.x python optimisers (API knowledge -- there is no reference-held vector for them: parity unpinned).
//   g = rescale * grad (+ wd * w where the rule folds weight decay into the gradient)
//   RMSPROP (non-centred): n = (1 - gamma1) g^2 + gamma1 n;  w -= lr g / sqrt(n + eps)                      [s1 = n;  p1 = gamma1]
//   ADAGRAD: h += g^2;  w -= lr (g / sqrt(h + eps) + wd w)                                                   [s1 = h]
//   ADADELTA: a = rho a + (1 - rho) g^2;  d = sqrt(b + eps) / sqrt(a + eps) g;  b = rho b + (1 - rho) d^2;  w -= d + wd w   [s1 = a, s2 = b; p1 = rho]
//   NAG: m = momentum m + g;  w -= lr (g + momentum m)                                                       [s1 = m;  p1 = momentum]
template <typename T, int KIND>
__global__ void opt_kernel(int64_t n, T* __restrict__ w, const T* __restrict__ g, T* __restrict__ s1, T* __restrict__ s2, T lr, T p1, T eps, T wd,
                           T rescale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T wi = w[i];
        if (KIND == MXF_OPT_RMSPROP) {
            const T gi = g[i] * rescale + wd * wi;
            const T ni = ((T)1 - p1) * gi * gi + p1 * s1[i];
            s1[i] = ni;
            w[i] = wi - lr * gi / sqrt(ni + eps);
        } else if (KIND == MXF_OPT_ADAGRAD) {
            const T gi = g[i] * rescale;
            const T hi = s1[i] + gi * gi;
            s1[i] = hi;
            w[i] = wi - lr * (gi / sqrt(hi + eps) + wd * wi);
        } else if (KIND == MXF_OPT_ADADELTA) {
            const T gi = g[i] * rescale;
            const T ai = p1 * s1[i] + ((T)1 - p1) * gi * gi;
            const T di = sqrt(s2[i] + eps) / sqrt(ai + eps) * gi;
            s1[i] = ai;
            s2[i] = p1 * s2[i] + ((T)1 - p1) * di * di;
            w[i] = wi - (di + wd * wi);
        } else {
            const T gi = g[i] * rescale + wd * wi;
            const T mi = p1 * s1[i] + gi;
            s1[i] = mi;
            w[i] = wi - lr * (gi + p1 * mi);
        }
    }
}

extern "C" int mxf_opt_step(mxf_handle h, int kind, int dtype, int64_t n, void* w, const void* g, void* s1, void* s2, double lr, double p1, double epsilon,
                            double wd, double rescale_grad, void* stream) {
    if (!h) return -1;
    if (n <= 0) return 0;
    if (!w || !g || !s1 || (kind == MXF_OPT_ADADELTA && !s2)) MXF_FAIL(h, -2, "mxf_opt_step: null argument");
    hipStream_t st = (hipStream_t)stream;
#define OPT_GO(K)                                                                                                                              \
    DISPATCH(h, dtype, "mxf_opt_step",                                                                                                          \
             hipLaunchKernelGGL((opt_kernel<float, K>), dim3(grid_for(n)), dim3(256), 0, st, n, (float*)w, (const float*)g, (float*)s1, (float*)s2, (float)lr, (float)p1, (float)epsilon, (float)wd, (float)rescale_grad), \
             hipLaunchKernelGGL((opt_kernel<double, K>), dim3(grid_for(n)), dim3(256), 0, st, n, (double*)w, (const double*)g, (double*)s1, (double*)s2, lr, p1, epsilon, wd, rescale_grad))
    switch (kind) {
        case MXF_OPT_RMSPROP: OPT_GO(MXF_OPT_RMSPROP);
        case MXF_OPT_ADAGRAD: OPT_GO(MXF_OPT_ADAGRAD);
        case MXF_OPT_ADADELTA: OPT_GO(MXF_OPT_ADADELTA);
        case MXF_OPT_NAG: OPT_GO(MXF_OPT_NAG);
    }
#undef OPT_GO
    MXF_FAIL(h, -2, "mxf_opt_step: unknown optimiser kind %d", kind);
}

extern "C" int mxf_coldot(mxf_handle h, int dtype, int S, int64_t M, int64_t N, const void* A, int64_t lda, int64_t strideS_A,
                          const void* B, int64_t ldb, int64_t strideS_B, void* out, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 g((unsigned)((N + 255) / 256), (unsigned)S);
    if ((int64_t)S * N < 16384 && M >= 64) {
        dim3 gs((unsigned)((N + 15) / 16), (unsigned)S);
        DISPATCH(h, dtype, "mxf_coldot",
                 hipLaunchKernelGGL((coldot_small_kernel<float>), gs, dim3(256), 0, st, M, N, (const float*)A, lda, strideS_A, (const float*)B, ldb, strideS_B, (float*)out),
                 hipLaunchKernelGGL((coldot_small_kernel<double>), gs, dim3(256), 0, st, M, N, (const double*)A, lda, strideS_A, (const double*)B, ldb, strideS_B, (double*)out));
    }
    DISPATCH(h, dtype, "mxf_coldot",
             hipLaunchKernelGGL((coldot_kernel<float>), g, dim3(256), 0, st, M, N, (const float*)A, lda, strideS_A, (const float*)B, ldb, strideS_B, (float*)out),
             hipLaunchKernelGGL((coldot_kernel<double>), g, dim3(256), 0, st, M, N, (const double*)A, lda, strideS_A, (const double*)B, ldb, strideS_B, (double*)out));
}

extern "C" int mxf_make_diagonal(mxf_handle h, int dtype, int64_t batch, int64_t n, const void* a, void* out, void* stream) {
    if (!h) return -1;
    if (batch <= 0 || n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = batch * n * n;
    DISPATCH(h, dtype, "mxf_make_diagonal",
             hipLaunchKernelGGL((make_diagonal_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, total, n, (const float*)a, (float*)out),
             hipLaunchKernelGGL((make_diagonal_kernel<double>), dim3(grid_for(total)), dim3(256), 0, st, total, n, (const double*)a, (double*)out));
}

extern "C" int mxf_diag_of(mxf_handle h, int dtype, int64_t batch, int64_t n, const void* g, void* out, void* stream) {
    if (!h) return -1;
    if (batch <= 0 || n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = batch * n;
    DISPATCH(h, dtype, "mxf_diag_of",
             hipLaunchKernelGGL((diag_of_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, total, n, (const float*)g, (float*)out),
             hipLaunchKernelGGL((diag_of_kernel<double>), dim3(grid_for(total)), dim3(256), 0, st, total, n, (const double*)g, (double*)out));
}

// Reverse mode of the stationary Gram build (gfx950):  dK -> dX, dX2, dlengthscale, dvariance in ONE
// streaming pass over dK (HBM-read bound: S*N*N2*sizeof(T) bytes), recomputing k(x,z) on the fly.
//
// This is what MXNet autograd does through stationary.py:92-106 + rbf.py:71-72 / matern.py:84-151 with
// ~10 materialised N^2 temporaries.  Mapping: lane <-> column (coalesced dK row reads), block loops over
// TR rows staged in LDS; column-side sums live in VGPRs, row-side sums use wavefront shuffle reductions.
#include "common.h"

namespace {

constexpr int TRB = 64;

template <typename T>
struct GramBwdArgs {
    const T* X; const T* X2; const T* ls; const T* var; const T* dK;
    T* dX; T* dX2; T* dls; T* dvar;
    int64_t N, N2, lddk;
    int64_t sX, sX2, sls, svar, sdK;
    int Q, ard, square;
};

// returns k and W = dk/d(r2) for unit variance (r2 in lengthscale-scaled coordinates)
template <typename T, int KIND>
__device__ __forceinline__ void cov_and_slope(T r2, T& k, T& w) {
    if (KIND == MXF_K_RBF) { k = exp((T)-0.5 * r2); w = (T)-0.5 * k; return; }
    const bool clipped = r2 < (T)1e-14;
    const T r = sqrt(clipped ? (T)1e-14 : r2);
    if (KIND == MXF_K_MATERN12) { k = exp(-r); w = clipped ? (T)0 : -k / ((T)2 * r); return; }
    if (KIND == MXF_K_MATERN32) {
        const T s3 = (T)1.7320508075688772, e = exp(-s3 * r);
        k = ((T)1 + s3 * r) * e; w = clipped ? (T)0 : (T)-1.5 * e; return;
    }
    const T s5 = (T)2.23606797749979, e = exp(-s5 * r);   // MATERN52 (matern.py:85-87: un-clipped r2 in the 5/3 term)
    k = ((T)1 + s5 * r + (T)(5.0 / 3.0) * r2) * e;
    w = clipped ? (T)(5.0 / 3.0) * e : (T)(-5.0 / 6.0) * ((T)1 + s5 * r) * e;
}

template <typename T, int QT, int KIND>
__global__ __launch_bounds__(256) void gram_bwd_kernel(GramBwdArgs<T> a) {
    __shared__ T xs[TRB * QT];
    __shared__ T red[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int s = blockIdx.z;
    const int64_t row0 = (int64_t)blockIdx.y * TRB;
    const int64_t col = (int64_t)blockIdx.x * 256 + tid;
    const int Q = a.Q;
    const T* __restrict__ X = a.X + (int64_t)s * a.sX;
    const T* __restrict__ X2 = a.X2 + (int64_t)s * a.sX2;
    const T* __restrict__ ls = a.ls + (int64_t)s * a.sls;
    const T variance = a.var[(int64_t)s * a.svar];
    const T* __restrict__ dK = a.dK + (int64_t)s * a.sdK;

    T il[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) il[q] = (q < Q) ? (T)1 / ls[a.ard ? q : 0] : (T)0;
    for (int i = tid; i < TRB * QT; i += 256) {
        const int r = i / QT, q = i % QT;
        const int64_t row = row0 + r;
        xs[i] = (row < a.N && q < Q) ? X[row * Q + q] / ls[a.ard ? q : 0] : (T)0;
    }
    const bool cvalid = col < a.N2;
    T z[QT], gz[QT], gl[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) { z[q] = (cvalid && q < Q) ? X2[col * Q + q] * il[q] : (T)0; gz[q] = 0; gl[q] = 0; }
    T gvar = 0;
    __syncthreads();

    const int64_t rmax = (a.N - row0) < TRB ? (a.N - row0) : TRB;
    for (int r = 0; r < rmax; ++r) {
        const int64_t row = row0 + r;
        const T g = cvalid ? dK[row * a.lddk + col] : (T)0;
        T d[QT], r2 = 0;
#pragma unroll
        for (int q = 0; q < QT; ++q) { d[q] = xs[r * QT + q] - z[q]; r2 = fma(d[q], d[q], r2); }
        T k, w;
        cov_and_slope<T, KIND>(r2, k, w);
        gvar = fma(g, k, gvar);
        const T W2 = (T)2 * g * w * variance;   // dL/d(r2) * 2
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const T t = W2 * d[q];              // dL/d(xs_q) in scaled coordinates
            gz[q] -= t;
            gl[q] = fma(-t, d[q], gl[q]);       // dL/dl_q * l_q
            if (a.dX) {
                T rs = wave_sum(t);
                if (lane == 0 && q < Q) atomic_add(a.dX + (int64_t)s * a.sX + row * Q + q, rs * il[q]);
            }
        }
    }
    // column side: in the square case both roles flow into dX
    T* dXc = a.square ? a.dX : a.dX2;
    const int64_t sXc = a.square ? a.sX : a.sX2;
    if (dXc && cvalid) {
#pragma unroll
        for (int q = 0; q < QT; ++q) if (q < Q) atomic_add(dXc + (int64_t)s * sXc + col * Q + q, gz[q] * il[q]);
    }
    if (a.dls) {
        if (a.ard) {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                T v = block_sum<T>(gl[q] * il[q], red);
                if (tid == 0 && q < Q) atomic_add(a.dls + (int64_t)s * a.sls + q, v);
            }
        } else {
            T v = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) v += gl[q];
            v = block_sum<T>(v * il[0], red);
            if (tid == 0) atomic_add(a.dls + (int64_t)s * a.sls, v);
        }
    }
    if (a.dvar) {
        T v = block_sum<T>(gvar, red);
        if (tid == 0) atomic_add(a.dvar + (int64_t)s * a.svar, v);
    }
}

template <typename T, int KIND>
int launch_bwd(mxf_ctx* h, const GramBwdArgs<T>& a, int S, hipStream_t st) {
    if (a.Q > 16) MXF_FAIL(h, -3, "mxf_gram_bwd: Q > 16 not supported");
    dim3 g((unsigned)((a.N2 + 255) / 256), (unsigned)((a.N + TRB - 1) / TRB), (unsigned)S);
    if (g.y > 65535u) MXF_FAIL(h, -3, "mxf_gram_bwd: N too large");
    if (a.Q <= 2) hipLaunchKernelGGL((gram_bwd_kernel<T, 2, KIND>), g, dim3(256), 0, st, a);
    else if (a.Q <= 4) hipLaunchKernelGGL((gram_bwd_kernel<T, 4, KIND>), g, dim3(256), 0, st, a);
    else if (a.Q <= 8) hipLaunchKernelGGL((gram_bwd_kernel<T, 8, KIND>), g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gram_bwd_kernel<T, 16, KIND>), g, dim3(256), 0, st, a);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

template <typename T>
int bwd_typed(mxf_ctx* h, int kind, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t sX, const void* X2, int64_t sX2,
              const void* ls, int ard, int64_t sls, const void* var, int64_t svar, const void* dK, int64_t lddk, int64_t sdK,
              void* dX, void* dX2, void* dls, void* dvar, hipStream_t st) {
    GramBwdArgs<T> a;
    a.square = (X2 == nullptr);
    a.X = (const T*)X; a.X2 = a.square ? (const T*)X : (const T*)X2; a.sX = sX; a.sX2 = a.square ? sX : sX2;
    a.ls = (const T*)ls; a.sls = sls; a.var = (const T*)var; a.svar = svar; a.dK = (const T*)dK; a.lddk = lddk; a.sdK = sdK;
    a.dX = (T*)dX; a.dX2 = (T*)dX2; a.dls = (T*)dls; a.dvar = (T*)dvar;
    a.N = N; a.N2 = a.square ? N : N2; a.Q = Q; a.ard = ard;
    switch (kind) {
        case MXF_K_RBF: return launch_bwd<T, MXF_K_RBF>(h, a, S, st);
        case MXF_K_MATERN12: return launch_bwd<T, MXF_K_MATERN12>(h, a, S, st);
        case MXF_K_MATERN32: return launch_bwd<T, MXF_K_MATERN32>(h, a, S, st);
        case MXF_K_MATERN52: return launch_bwd<T, MXF_K_MATERN52>(h, a, S, st);
    }
    MXF_FAIL(h, -2, "mxf_gram_bwd: kind %d has no stationary reverse mode", kind);
}

}  // namespace

int mxf_gram_bwd_internal(mxf_ctx* h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t sX,
                          const void* X2, int64_t sX2, const void* ls, int ard, int64_t sls, const void* var, int64_t svar,
                          const void* dK, int64_t lddk, int64_t sdK, void* dX, void* dX2, void* dls, void* dvar, hipStream_t st) {
    if (S <= 0 || N <= 0 || (X2 && N2 <= 0)) return 0;
    if (dtype == MXF_F32) return bwd_typed<float>(h, kind, S, N, N2, Q, X, sX, X2, sX2, ls, ard, sls, var, svar, dK, lddk, sdK, dX, dX2, dls, dvar, st);
    if (dtype == MXF_F64) return bwd_typed<double>(h, kind, S, N, N2, Q, X, sX, X2, sX2, ls, ard, sls, var, svar, dK, lddk, sdK, dX, dX2, dls, dvar, st);
    MXF_FAIL(h, -2, "mxf_gram_bwd: bad dtype %d", dtype);
}

extern "C" int mxf_gram_bwd(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q,
                            const void* X, int64_t strideS_X, const void* X2, int64_t strideS_X2,
                            const void* lengthscale, int ard, int64_t strideS_ls,
                            const void* variance, int64_t strideS_var,
                            const void* dK, int64_t lddk, int64_t strideS_dK,
                            void* dX, void* dX2, void* dls, void* dvar, void* stream) {
    if (!h) return -1;
    if (S < 0 || N < 0 || N2 < 0 || Q <= 0) MXF_FAIL(h, -2, "mxf_gram_bwd: bad shape");
    if (!X || !lengthscale || !variance || !dK) MXF_FAIL(h, -2, "mxf_gram_bwd: null input");
    return mxf_gram_bwd_internal(h, kind, dtype, S, N, N2, Q, X, strideS_X, X2, strideS_X2, lengthscale, ard, strideS_ls, variance,
                                 strideS_var, dK, lddk, strideS_dK, dX, dX2, dls, dvar, (hipStream_t)stream);
}

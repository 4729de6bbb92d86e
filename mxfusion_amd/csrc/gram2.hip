// K = k1(X, X2) (+ | *) k2(X, X2) for two stationary kernels on the same inputs, ONE pass and ONE write -- AddKernel / MultiplyKernel
// (kernels/add_kernel.py:44-68, multiply_kernel.py:44-67; SURVEY section 8 f1: "AddKernel fusion in the Gram epilogue").  The reference
// materialises each sub-kernel's Gram and adds them (three N x N2 passes for two kernels); mxf_gram's accumulate modes need a write plus a
// read-modify-write.  Here both covariances come from the SAME coordinate differences: r1^2 = sum_q d_q^2 (c1 / l1_q)^2 and r2^2 with the
// second kernel's length-scales -- differences first, scaling after (the accurate order).
// Lane <-> VEC consecutive columns (16-byte stores), one-wave workgroups walk TR rows; the row's coordinates are wave-uniform scalar loads.
#include "common.h"
#include "internal.h"

namespace {

template <typename T> __device__ __forceinline__ T g2_exp2_neg(T x);
template <> __device__ __forceinline__ float g2_exp2_neg<float>(float x) { return __builtin_amdgcn_exp2f(-x); }
template <> __device__ __forceinline__ double g2_exp2_neg<double>(double x) { return mxf_exp2_neg_f64(x); }
template <typename T> __device__ __forceinline__ T g2_exp_nonpos(T x);
template <> __device__ __forceinline__ float g2_exp_nonpos<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double g2_exp_nonpos<double>(double x) { return mxf_exp_nonpos_f64(x); }
template <typename T> __device__ __forceinline__ T g2_sqrt(T x);
template <> __device__ __forceinline__ float g2_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double g2_sqrt<double>(double x) { return sqrt(x); }

// covariance from the scaled squared distance (RBF: scaled so that k = v 2^-red; Matern: red = r^2, clipped at 1e-14 as matern.py:84-88 does)
template <typename T>
__device__ __forceinline__ T g2_cov(int kind, T red, T v) {
    if (kind == MXF_K_RBF) return v * g2_exp2_neg<T>(red);
    const T r0 = g2_sqrt<T>(red < (T)1e-14 ? (T)1e-14 : red);
    if (kind == MXF_K_MATERN12) return v * g2_exp_nonpos<T>(-r0);
    if (kind == MXF_K_MATERN32) { const T r = (T)1.7320508075688772 * r0; return v * ((T)1 + r) * g2_exp_nonpos<T>(-r); }
    const T r = (T)2.23606797749979 * r0;                                   // MATERN52: un-clipped r^2 in the 5/3 r^2 term (matern.py:87)
    return v * ((T)1 + r + (T)(5.0 / 3.0) * red) * g2_exp_nonpos<T>(-r);
}

struct G2Args {
    const void* X; const void* X2; const void* ls1; const void* var1; const void* ls2; const void* var2; const void* dadd; void* K;
    int64_t N, N2, ldk, sX, sX2, sls1, svar1, sls2, svar2, sdadd, sK;
    int Q, ard1, ard2, kind1, kind2, op, square;
    double jitter;
};

constexpr int G2_TR = 16;
template <typename T, int QT>
__global__ __launch_bounds__(64) void gram2_kernel(G2Args a) {
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    const int lane = threadIdx.x, s = blockIdx.z;
    const int64_t col0 = ((int64_t)blockIdx.x * 64 + lane) * VEC, row0 = (int64_t)blockIdx.y * G2_TR;
    const T* X = (const T*)a.X + (int64_t)s * a.sX;
    const T* X2 = (const T*)a.X2 + (int64_t)s * a.sX2;
    const T* l1 = (const T*)a.ls1 + (int64_t)s * a.sls1;
    const T* l2 = (const T*)a.ls2 + (int64_t)s * a.sls2;
    const T v1 = ((const T*)a.var1)[(int64_t)s * a.svar1], v2 = ((const T*)a.var2)[(int64_t)s * a.svar2];
    const T dadd = a.square ? ((a.dadd ? ((const T*)a.dadd)[(int64_t)s * a.sdadd] : (T)0) + (T)a.jitter) : (T)0;
    const T c1 = a.kind1 == MXF_K_RBF ? (T)0.84932180028801904272 : (T)1, c2 = a.kind2 == MXF_K_RBF ? (T)0.84932180028801904272 : (T)1;
    T s1[QT], s2[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const T m1 = q < a.Q ? c1 / l1[a.ard1 ? q : 0] : (T)0, m2 = q < a.Q ? c2 / l2[a.ard2 ? q : 0] : (T)0;
        s1[q] = m1 * m1; s2[q] = m2 * m2;
    }
    T z[VEC][QT];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int q = 0; q < QT; ++q) z[v][q] = (col0 + v < a.N2 && q < a.Q) ? X2[(col0 + v) * a.Q + q] : (T)0;
    if (col0 >= a.N2) return;
    T* K = (T*)a.K + (int64_t)s * a.sK;
    const bool vec_ok = (col0 + VEC <= a.N2) && (a.ldk % VEC == 0) && (((uintptr_t)K) % 16 == 0);
    for (int r = 0; r < G2_TR; ++r) {
        const int64_t row = row0 + r;
        if (row >= a.N) break;
        T x[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) x[q] = q < a.Q ? X[row * a.Q + q] : (T)0;          // wave-uniform
        T kv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            T r1 = 0, r2 = 0;
#pragma unroll
            for (int q = 0; q < QT; ++q) { const T d = x[q] - z[v][q], d2 = d * d; r1 = fma(d2, s1[q], r1); r2 = fma(d2, s2[q], r2); }
            const T k1 = g2_cov<T>(a.kind1, r1, v1), k2 = g2_cov<T>(a.kind2, r2, v2);
            kv[v] = a.op == MXF_ACC_MUL ? k1 * k2 : k1 + k2;
            if (a.square && col0 + v == row) kv[v] += dadd;
        }
        T* dst = K + row * a.ldk + col0;
        if (vec_ok) {
            V out;
            T* po = reinterpret_cast<T*>(&out);
#pragma unroll
            for (int v = 0; v < VEC; ++v) po[v] = kv[v];
            *reinterpret_cast<V*>(dst) = out;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) if (col0 + v < a.N2) dst[v] = kv[v];
        }
    }
}

}  // namespace

extern "C" int mxf_gram2(mxf_handle h, int kind1, int kind2, int op, int dtype, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t strideS_X,
                         const void* X2, int64_t strideS_X2, const void* lengthscale1, int ard1, int64_t strideS_ls1, const void* variance1,
                         int64_t strideS_var1, const void* lengthscale2, int ard2, int64_t strideS_ls2, const void* variance2, int64_t strideS_var2,
                         const void* diag_add, int64_t strideS_diag, double jitter, void* K_out, int64_t ldk, int64_t strideS_K, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N < 0 || N2 < 0 || Q <= 0) MXF_FAIL(h, -2, "mxf_gram2: bad shape");
    const int64_t n2 = X2 ? N2 : N;
    if (N == 0 || n2 == 0) return 0;
    if (!X || !K_out || !lengthscale1 || !variance1 || !lengthscale2 || !variance2) MXF_FAIL(h, -2, "mxf_gram2: null argument");
    if (kind1 < MXF_K_RBF || kind1 > MXF_K_MATERN52 || kind2 < MXF_K_RBF || kind2 > MXF_K_MATERN52) MXF_FAIL(h, -2, "mxf_gram2: stationary kernels only");
    if (op != MXF_ACC_ADD && op != MXF_ACC_MUL) MXF_FAIL(h, -2, "mxf_gram2: op must be MXF_ACC_ADD or MXF_ACC_MUL");
    if (Q > 16) MXF_FAIL(h, -3, "mxf_gram2: Q > 16 not supported (combine two mxf_gram calls)");
    if (ldk < n2) MXF_FAIL(h, -2, "mxf_gram2: ldk < N2");
    if (S > 65535) MXF_FAIL(h, -3, "mxf_gram2: S too large");
    if (dtype != MXF_F32 && dtype != MXF_F64) MXF_FAIL(h, -2, "mxf_gram2: bad dtype %d", dtype);
    G2Args a;
    a.X = X; a.square = X2 == nullptr; a.X2 = a.square ? X : X2; a.sX = strideS_X; a.sX2 = a.square ? strideS_X : strideS_X2;
    a.ls1 = lengthscale1; a.var1 = variance1; a.ls2 = lengthscale2; a.var2 = variance2; a.dadd = diag_add; a.K = K_out;
    a.N = N; a.N2 = n2; a.ldk = ldk; a.sls1 = strideS_ls1; a.svar1 = strideS_var1; a.sls2 = strideS_ls2; a.svar2 = strideS_var2;
    a.sdadd = strideS_diag; a.sK = strideS_K; a.Q = Q; a.ard1 = ard1; a.ard2 = ard2; a.kind1 = kind1; a.kind2 = kind2; a.op = op; a.jitter = jitter;
    const int vec = dtype == MXF_F32 ? 4 : 2;
    const int64_t cb = (n2 + 64 * vec - 1) / (64 * vec), rb = (N + G2_TR - 1) / G2_TR;
    if (rb > 65535) {
        MXF_FAIL(h, -3, "mxf_gram2: N too large for one launch (%lld rows)", (long long)N);
    }
    dim3 grid((unsigned)cb, (unsigned)rb, (unsigned)S);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32) { if (Q <= 8) hipLaunchKernelGGL((gram2_kernel<float, 8>), grid, dim3(64), 0, st, a); else hipLaunchKernelGGL((gram2_kernel<float, 16>), grid, dim3(64), 0, st, a); }
    else { if (Q <= 8) hipLaunchKernelGGL((gram2_kernel<double, 8>), grid, dim3(64), 0, st, a); else hipLaunchKernelGGL((gram2_kernel<double, 16>), grid, dim3(64), 0, st, a); }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

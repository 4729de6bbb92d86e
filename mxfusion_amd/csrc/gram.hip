// Gram-matrix build for gfx950 (MI355X): one fused pass  X, X2, lengthscale, variance -> K.
//
// Replaces StationaryKernel._compute_R2 (kernels/stationary.py:74-107: syrk/gemm2 + 3 broadcast passes)
// and RBF/Matern._compute_K (rbf.py:71-72, matern.py:84-151: 2-6 more N^2 passes) of the reference.
//
// Roofline: HBM-WRITE bound.  Algorithmic bytes = S*N*N2*sizeof(T) written (+ (N+N2)*Q read).
// Layout / mapping (wave64):
//   * a wave owns a strip of 64*VEC consecutive columns (VEC = 16 B / sizeof(T)): lane l keeps the
//     VEC pre-scaled z-vectors of its columns in VGPRs for the whole row loop;
//   * the block (4 waves side by side = 1024 f32 / 512 f64 columns) stages TR pre-scaled rows of X in LDS;
//     every lane reads a row with broadcast ds_read_b128 (same address in all lanes: conflict free);
//   * each lane produces VEC outputs per row and stores them with ONE 16-byte store, so a wave writes
//     1 KiB of one output row per instruction (full-line, fully coalesced), non-temporal (written once,
//     never re-read by this kernel).
#include "common.h"

namespace {

constexpr int TR = 64;   // rows of X per block

template <typename T>
struct GramArgs {
    const T* X; const T* X2; const T* ls; const T* var; const T* dadd;
    T* K;
    int64_t N, N2, ldk;
    int64_t sX, sX2, sls, svar, sdadd, sK;
    int Q, ard, square, mode, vecst;
    T jitter;
};

template <typename T> __device__ __forceinline__ T fast_exp2_neg(T x);   // 2^(-x), x >= 0
template <> __device__ __forceinline__ float fast_exp2_neg<float>(float x) { return __builtin_amdgcn_exp2f(-x); }
template <> __device__ __forceinline__ double fast_exp2_neg<double>(double x) { return exp2(-x); }

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }

// coordinate pre-scale so that the RBF epilogue is a bare exp2:  exp(-r2/2) = 2^-(c^2 r2), c^2 = log2(e)/2
template <typename T, int KIND> __device__ __forceinline__ T coord_scale() {
    return KIND == MXF_K_RBF ? (T)0.84932180028801904272 /* sqrt(0.5*log2(e)) */ : (T)1;
}

// value of the covariance from the reduced quantity (scaled r2, or the dot product for LINEAR)
template <typename T, int KIND> __device__ __forceinline__ T cov_from(T red, T variance) {
    if (KIND == MXF_K_RBF) return variance * fast_exp2_neg<T>(red);
    if (KIND == MXF_K_MATERN12) { T r = t_sqrt<T>(red < (T)1e-14 ? (T)1e-14 : red); return variance * t_exp<T>(-r); }
    if (KIND == MXF_K_MATERN32) {
        T r = (T)1.7320508075688772 * t_sqrt<T>(red < (T)1e-14 ? (T)1e-14 : red);
        return variance * ((T)1 + r) * t_exp<T>(-r);
    }
    if (KIND == MXF_K_MATERN52) {   // matern.py:85-87: clipped r in the linear/exp terms, UN-clipped r2 in 5/3 r2
        T r = (T)2.23606797749979 * t_sqrt<T>(red < (T)1e-14 ? (T)1e-14 : red);
        return variance * ((T)1 + r + (T)(5.0 / 3.0) * red) * t_exp<T>(-r);
    }
    return red;   // LINEAR: the dot product itself
}

template <typename T, int QT, int KIND>
__global__ __launch_bounds__(256) void gram_kernel(GramArgs<T> a) {
    const int MODE = a.mode;
    const bool VECST = a.vecst;
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    __shared__ __attribute__((aligned(16))) T xs[TR * QT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.z;
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (64 * VEC) + (int64_t)lane * VEC;

    const T* __restrict__ X = a.X + (int64_t)s * a.sX;
    const T* __restrict__ X2 = a.X2 + (int64_t)s * a.sX2;
    const T* __restrict__ ls = a.ls + (int64_t)s * a.sls;
    const T variance = (KIND == MXF_K_LINEAR) ? (T)1 : a.var[(int64_t)s * a.svar];
    T* __restrict__ K = a.K + (int64_t)s * a.sK;
    const int Q = a.Q;

    // per-dimension multiplier: 1/l_q (stationary) or sqrt(v_q) (linear), folded with the exp2 constant
    T mult[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        if (KIND == MXF_K_BIAS || KIND == MXF_K_WHITE) { mult[q] = (T)0; continue; }
        T l = (q < Q) ? ls[a.ard ? q : 0] : (T)1;
        if (KIND == MXF_K_LINEAR) mult[q] = (q < Q) ? t_sqrt<T>(l) : (T)0;
        else mult[q] = (q < Q) ? coord_scale<T, KIND>() / l : (T)0;
    }

    if (KIND != MXF_K_BIAS && KIND != MXF_K_WHITE) {
        for (int i = tid; i < TR * QT; i += 256) {
            const int r = i / QT, q = i % QT;
            const int64_t row = row0 + r;
            T l = (q < Q) ? ls[a.ard ? q : 0] : (T)1;
            T m = (KIND == MXF_K_LINEAR) ? t_sqrt<T>(l) : coord_scale<T, KIND>() / l;
            xs[i] = (row < a.N && q < Q) ? X[row * Q + q] * m : (T)0;
        }
    }
    T z[VEC][QT];
    if (KIND != MXF_K_BIAS && KIND != MXF_K_WHITE) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t col = col0 + v;
#pragma unroll
            for (int q = 0; q < QT; ++q) z[v][q] = (col < a.N2 && q < Q) ? X2[col * Q + q] * mult[q] : (T)0;
        }
    }
    __syncthreads();
    if (col0 >= a.N2) return;

    const T dadd = a.square ? ((a.dadd ? a.dadd[(int64_t)s * a.sdadd] : (T)0) + a.jitter) : (T)0;
    const bool full = VECST && (col0 + VEC <= a.N2);
    const int64_t rmax = (a.N - row0) < TR ? (a.N - row0) : TR;

#pragma unroll 2
    for (int r = 0; r < rmax; ++r) {
        const int64_t row = row0 + r;
        T kv[VEC];
        if (KIND == MXF_K_BIAS) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) kv[v] = variance;
        } else if (KIND == MXF_K_WHITE) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) kv[v] = (a.square && (col0 + v == row)) ? variance : (T)0;
        } else {
            T x[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) x[q] = xs[r * QT + q];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                T acc = 0;
                if (KIND == MXF_K_LINEAR) {
#pragma unroll
                    for (int q = 0; q < QT; ++q) acc = fma(x[q], z[v][q], acc);
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) { T d = x[q] - z[v][q]; acc = fma(d, d, acc); }
                }
                kv[v] = cov_from<T, KIND>(acc, variance);
            }
        }
        if (dadd != (T)0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) if (col0 + v == row) kv[v] += dadd;
        }
        T* dst = K + row * a.ldk + col0;
        if (full) {
            V out;
            if (MODE != MXF_WRITE) {
                V old = *reinterpret_cast<const V*>(dst);
                T* o = reinterpret_cast<T*>(&old);
#pragma unroll
                for (int v = 0; v < VEC; ++v) kv[v] = (MODE == MXF_ACC_ADD) ? o[v] + kv[v] : o[v] * kv[v];
            }
            T* po = reinterpret_cast<T*>(&out);
#pragma unroll
            for (int v = 0; v < VEC; ++v) po[v] = kv[v];
            if (MODE == MXF_WRITE) __builtin_nontemporal_store(out, reinterpret_cast<V*>(dst));
            else *reinterpret_cast<V*>(dst) = out;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (col0 + v < a.N2) {
                    T val = kv[v];
                    if (MODE == MXF_ACC_ADD) val = dst[v] + val;
                    if (MODE == MXF_ACC_MUL) val = dst[v] * val;
                    dst[v] = val;
                }
            }
        }
    }
}

// generic fallback for Q > 16: one output per thread, q-loop over global memory
template <typename T, int KIND>
__global__ __launch_bounds__(256) void gram_generic_kernel(GramArgs<T> a) {
    const int MODE = a.mode;
    const int s = blockIdx.z;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    if (col >= a.N2) return;
    const T* X = a.X + (int64_t)s * a.sX + row * a.Q;
    const T* X2 = a.X2 + (int64_t)s * a.sX2 + col * a.Q;
    const T* ls = a.ls + (int64_t)s * a.sls;
    const T variance = (KIND == MXF_K_LINEAR) ? (T)1 : a.var[(int64_t)s * a.svar];
    T acc = 0;
    if (KIND == MXF_K_LINEAR) {
        for (int q = 0; q < a.Q; ++q) acc = fma(X[q] * ls[a.ard ? q : 0], X2[q], acc);
    } else if (KIND != MXF_K_BIAS && KIND != MXF_K_WHITE) {
        for (int q = 0; q < a.Q; ++q) { T d = (X[q] - X2[q]) * (coord_scale<T, KIND>() / ls[a.ard ? q : 0]); acc = fma(d, d, acc); }
    }
    T k;
    if (KIND == MXF_K_BIAS) k = variance;
    else if (KIND == MXF_K_WHITE) k = (a.square && row == col) ? variance : (T)0;
    else k = cov_from<T, KIND>(acc, variance);
    if (a.square && row == col) k += (a.dadd ? a.dadd[(int64_t)s * a.sdadd] : (T)0) + a.jitter;
    T* dst = a.K + (int64_t)s * a.sK + row * a.ldk + col;
    if (MODE == MXF_ACC_ADD) k = *dst + k;
    if (MODE == MXF_ACC_MUL) k = *dst * k;
    *dst = k;
}

template <typename T, int KIND>
int launch_kind(mxf_ctx* h, GramArgs<T> a, int S, int mode, hipStream_t st) {
    constexpr int VEC = Vec16<T>::n;
    if (mode < 0 || mode > 2) MXF_FAIL(h, -2, "mxf_gram: bad mode %d", mode);
    a.mode = mode;
    if (a.Q > 16) {
        dim3 g((unsigned)((a.N2 + 255) / 256), (unsigned)a.N, (unsigned)S);
        if (a.N > 65535) MXF_FAIL(h, -3, "mxf_gram: Q>16 fallback supports N<=65535");
        hipLaunchKernelGGL((gram_generic_kernel<T, KIND>), g, dim3(256), 0, st, a);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    a.vecst = (a.ldk % VEC == 0) && (a.sK % VEC == 0) && (((uintptr_t)a.K) % 16 == 0);
    dim3 g((unsigned)((a.N2 + 4 * 64 * VEC - 1) / (4 * 64 * VEC)), (unsigned)((a.N + TR - 1) / TR), (unsigned)S);
    if (g.y > 65535u) MXF_FAIL(h, -3, "mxf_gram: N too large for one launch");
#define GO(QT) hipLaunchKernelGGL((gram_kernel<T, QT, KIND>), g, dim3(256), 0, st, a)
    if (KIND == MXF_K_BIAS || KIND == MXF_K_WHITE) GO(2);
    else if (a.Q <= 2) GO(2);
    else if (a.Q <= 4) GO(4);
    else if (a.Q <= 8) GO(8);
    else GO(16);
#undef GO
    MXF_LAUNCH_CHECK(h);
    return 0;
}

template <typename T>
int gram_typed(mxf_ctx* h, int kind, int S, int64_t N, int64_t N2, int Q,
               const void* X, int64_t sX, const void* X2, int64_t sX2, const void* ls, int ard, int64_t sls,
               const void* var, int64_t svar, const void* dadd, int64_t sdadd, double jitter, int mode,
               void* K, int64_t ldk, int64_t sK, hipStream_t st) {
    GramArgs<T> a;
    a.X = (const T*)X; a.square = (X2 == nullptr);
    a.X2 = a.square ? (const T*)X : (const T*)X2;
    a.sX = sX; a.sX2 = a.square ? sX : sX2;
    a.ls = (const T*)ls; a.sls = sls; a.var = (const T*)var; a.svar = svar;
    a.dadd = (const T*)dadd; a.sdadd = sdadd; a.jitter = (T)jitter;
    a.K = (T*)K; a.N = N; a.N2 = a.square ? N : N2; a.ldk = ldk; a.sK = sK; a.Q = Q; a.ard = ard;
    switch (kind) {
        case MXF_K_RBF: return launch_kind<T, MXF_K_RBF>(h, a, S, mode, st);
        case MXF_K_MATERN12: return launch_kind<T, MXF_K_MATERN12>(h, a, S, mode, st);
        case MXF_K_MATERN32: return launch_kind<T, MXF_K_MATERN32>(h, a, S, mode, st);
        case MXF_K_MATERN52: return launch_kind<T, MXF_K_MATERN52>(h, a, S, mode, st);
        case MXF_K_LINEAR: return launch_kind<T, MXF_K_LINEAR>(h, a, S, mode, st);
        case MXF_K_BIAS: return launch_kind<T, MXF_K_BIAS>(h, a, S, mode, st);
        case MXF_K_WHITE: return launch_kind<T, MXF_K_WHITE>(h, a, S, mode, st);
    }
    MXF_FAIL(h, -2, "mxf_gram: unknown kernel kind %d", kind);
}

}  // namespace

extern "C" int mxf_gram(mxf_handle h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q,
                        const void* X, int64_t strideS_X, const void* X2, int64_t strideS_X2,
                        const void* lengthscale, int ard, int64_t strideS_ls,
                        const void* variance, int64_t strideS_var,
                        const void* diag_add, int64_t strideS_diag, double jitter, int mode,
                        void* K_out, int64_t ldk, int64_t strideS_K, void* stream) {
    if (!h) return -1;
    if (S <= 0 || N < 0 || N2 < 0 || Q <= 0) MXF_FAIL(h, -2, "mxf_gram: bad shape S=%d N=%lld N2=%lld Q=%d", S, (long long)N, (long long)N2, Q);
    const int64_t n2 = X2 ? N2 : N;
    if (N == 0 || n2 == 0) return 0;
    if (!X || !K_out) MXF_FAIL(h, -2, "mxf_gram: null X or K_out");
    if (kind != MXF_K_BIAS && kind != MXF_K_WHITE && !lengthscale) MXF_FAIL(h, -2, "mxf_gram: null lengthscale/variances");
    if (kind != MXF_K_LINEAR && !variance) MXF_FAIL(h, -2, "mxf_gram: null variance");
    if (ldk < n2) MXF_FAIL(h, -2, "mxf_gram: ldk < N2");
    if (S > 65535) MXF_FAIL(h, -3, "mxf_gram: S too large");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MXF_F32)
        return gram_typed<float>(h, kind, S, N, N2, Q, X, strideS_X, X2, strideS_X2, lengthscale, ard, strideS_ls,
                                 variance, strideS_var, diag_add, strideS_diag, jitter, mode, K_out, ldk, strideS_K, st);
    if (dtype == MXF_F64)
        return gram_typed<double>(h, kind, S, N, N2, Q, X, strideS_X, X2, strideS_X2, lengthscale, ard, strideS_ls,
                                  variance, strideS_var, diag_add, strideS_diag, jitter, mode, K_out, ldk, strideS_K, st);
    MXF_FAIL(h, -2, "mxf_gram: bad dtype %d", dtype);
}

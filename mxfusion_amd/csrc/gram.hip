// Gram-matrix build for gfx950 (MI355X): one fused pass  X, X2, lengthscale, variance -> K.
//
// Replaces StationaryKernel._compute_R2 (kernels/stationary.py:74-107: syrk/gemm2 + 3 broadcast passes)
// and RBF/Matern._compute_K (rbf.py:71-72, matern.py:84-151: 2-6 more N^2 passes) of the reference.
//
// Roofline: HBM-WRITE bound.  Algorithmic bytes = S*N*N2*sizeof(T) written (+ (N+N2)*Q read).
// Layout / mapping (wave64):
//   * a wave owns a strip of 64*VEC consecutive columns (VEC = 16 B / sizeof(T)): lane l keeps the
//     VEC pre-scaled z-vectors of its columns in VGPRs for the whole row loop;
//   * a workgroup is ONE wave (64 threads) covering 16 rows of its strip; the pre-scaled x row is the same for all lanes and comes
//     through scalar loads (s_load_dwordx8/x16 into SGPRs, used directly as VOP3P sources): no LDS, no barrier;
//   * each lane produces VEC outputs per row and stores them with ONE 16-byte store, so a wave writes
//     1 KiB of one output row per instruction (full-line, fully coalesced), non-temporal (written once,
//     never re-read by this kernel).
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TR = 64;   // rows of X per block

template <typename T>
struct GramArgs {
    const T* X; const T* X2; const T* ls; const T* var; const T* dadd;
    const T* Xs; const T* Zs; int64_t sXs, sZs;   // pre-scaled, zero-padded coordinates [S|1][pad][QT] (prescale_kernel)
    T* K;
    int64_t N, N2, ldk;
    int64_t sX, sX2, sls, svar, sdadd, sK;
    int Q, ard, square, mode, vecst, tr, nt;
    unsigned ncb;          // column blocks (grid.x = ncb * row blocks, column block fastest)
    T jitter;
};

template <typename T> __device__ __forceinline__ T fast_exp2_neg(T x);   // 2^(-x), x >= 0
template <> __device__ __forceinline__ float fast_exp2_neg<float>(float x) { return __builtin_amdgcn_exp2f(-x); }
template <> __device__ __forceinline__ double fast_exp2_neg<double>(double x) { return mxf_exp2_neg_f64(x); }

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return mxf_exp_nonpos_f64(x); }   // only called with x <= 0

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

// 2^(t / 64) for float64 from a 64-entry table of 2^(j / 64) (LDS) and a degree-5 polynomial on |g| <= 1/2 (g in 64ths): 11 VALU
// operations + one ds_read_b64 instead of the 18 of the degree-13 form (truncation (ln2/128)^6 / 6! = 3.5e-17; result within ~1 ulp).
__device__ __forceinline__ double exp2_64ths_tab(double t, const double* __restrict__ tab) {
    const double m = __builtin_rint(t), g = t - m;
    double p = 1.241784370171692541187e-12;
    p = fma(p, g, 5.732851688640402055966e-10); p = fma(p, g, 2.117313715546477506757e-7); p = fma(p, g, 0.00005864904955056169734706);
    p = fma(p, g, 0.01083042469624914545964); p = fma(p, g, 1.0);
    const int mi = (int)m;
    return ldexp(p * tab[mi & 63], mi >> 6);
}

// NW = waves per workgroup.  The x rows are read through wave-uniform (scalar) loads straight from the pre-scaled copy: no LDS, no
// barrier, and with NW = 1 every wave is its own workgroup, dispatched and retired independently (measured on MI355X, f32 RBF N=65536:
// 4-wave blocks with an LDS x tile 5.67 TB/s, 4-wave blocks with scalar x 5.86, single-wave blocks with scalar x 6.52 TB/s --
// tests/probes/gram_variants.hip).
// FAST: plain overwrite (MXF_WRITE), 16-byte-aligned rows and N2 a multiple of VEC -- every store is one full 16-byte store and the
// accumulate / ragged-edge code (and its registers: 86 -> <= 64 VGPRs, i.e. 8 waves per SIMD) is compiled out.
template <typename T, int QT, int KIND, int NW, bool FAST>
__global__ __launch_bounds__(NW * 64) void gram_kernel(GramArgs<T> a, const T* __restrict__ Xs_all, const T* __restrict__ Zs_all,
                                                       T* __restrict__ K_all) {
    // (the three array pointers are separate __restrict__ kernel arguments: only then may the compiler read the x rows with SCALAR
    //  loads -- a pointer inside the by-value struct could alias the output and would be fetched with per-lane vector loads)
    const int MODE = FAST ? (int)MXF_WRITE : a.mode;
    const bool VECST = FAST ? true : (bool)a.vecst;
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;

    const int tid = threadIdx.x, lane = tid & 63, wave = NW == 1 ? 0 : tid >> 6;
    const int s = blockIdx.z;
    const int TRr = a.tr;                       // rows per block (<= TR)
    const int64_t row0 = (int64_t)(blockIdx.x / a.ncb) * TRr;
    const int64_t col0 = ((int64_t)(blockIdx.x % a.ncb) * NW + wave) * (64 * VEC) + (int64_t)lane * VEC;
    if (col0 >= a.N2 || row0 >= a.N) return;

    const T variance = (KIND == MXF_K_LINEAR) ? (T)1 : a.var[(int64_t)s * a.svar];
    T* __restrict__ K = K_all + (int64_t)s * a.sK;

    // prologue: unguarded 16-byte copies of the PRE-SCALED, zero-padded coordinates (prescale_kernel): the VEC z-vectors of this
    // lane's columns into VGPRs.
    T z[VEC][QT];
    const T* __restrict__ Xrows = nullptr;
    if (KIND != MXF_K_BIAS && KIND != MXF_K_WHITE) {
        Xrows = Xs_all + (int64_t)s * a.sXs + row0 * QT;       // wave-uniform
        const T* __restrict__ Zs = Zs_all + (int64_t)s * a.sZs + col0 * QT;
#pragma unroll
        for (int i = 0; i < VEC * QT; i += VEC)
            *reinterpret_cast<V*>(&z[0][0] + i) = *reinterpret_cast<const V*>(Zs + i);
    }

    const T dadd = a.square ? ((a.dadd ? a.dadd[(int64_t)s * a.sdadd] : (T)0) + a.jitter) : (T)0;
    const bool full = FAST ? true : (VECST && (col0 + VEC <= a.N2));
    // (an interleaved row assignment -- the NB workgroups that run side by side own every NB-th row of a band, so that the rows being
    //  written at any moment are consecutive -- was measured: 6.4 -> 4.1-5.4 TB/s; rows per workgroup stay consecutive)

    // one row: kv = covariances of (row, col0 .. col0+VEC-1), then the store.  DIAG (the "+ noise / jitter on the diagonal" and the
    // WHITE kernel) is a compile-time flag: only the few waves whose tile touches the diagonal run the variant with the compares.
    // the "+ noise / jitter on the diagonal" and the WHITE kernel touch one element per row: a SCALAR test (is this row inside the
    // wave's column range?) guards the per-lane compares, so the row loop carries no vector compare for them
    const int64_t wcol0 = ((int64_t)(blockIdx.x % a.ncb) * NW + wave) * (64 * VEC);      // first column of the wave (wave-uniform)
    const bool diag_possible = __builtin_amdgcn_readfirstlane((int)(a.square && (KIND == MXF_K_WHITE || dadd != (T)0))) != 0;
    auto do_row = [&](int r) {
        const int64_t row = row0 + r;
        const bool DIAG = diag_possible && (uint64_t)(row - wcol0) < (uint64_t)(64 * VEC);
        T kv[VEC];
        if (KIND == MXF_K_BIAS) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) kv[v] = variance;
        } else if (KIND == MXF_K_WHITE) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) kv[v] = (DIAG && (col0 + v == row)) ? variance : (T)0;
        } else {
            T x[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) x[q] = Xrows[r * QT + q];      // s_load: the address is the same in every lane
            if constexpr (sizeof(T) == 4 && KIND != MXF_K_LINEAR) {
                // float: two outputs per v_pk_add_f32 / v_pk_fma_f32 (halves the VALU issue slots of the distance loop)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int p = 0; p < VEC / 2; ++p) {
                    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const f32x2 xx = {(float)x[q], (float)x[q]};
                        const f32x2 zz = {(float)z[2 * p][q], (float)z[2 * p + 1][q]};
                        const f32x2 d = xx - zz;
                        acc2 = __builtin_elementwise_fma(d, d, acc2);
                    }
                    kv[2 * p] = cov_from<T, KIND>((T)acc2.x, variance);
                    kv[2 * p + 1] = cov_from<T, KIND>((T)acc2.y, variance);
                }
            } else
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
        if (DIAG) {
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
            if (MODE == MXF_WRITE && (FAST || a.nt)) __builtin_nontemporal_store(out, reinterpret_cast<V*>(dst));
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
    };
    // number of this workgroup's rows that exist (wave-uniform)
    const int rmax = (a.N - row0) < TRr ? (int)(a.N - row0) : TRr;
#pragma unroll 2
    for (int r = 0; r < rmax; ++r) do_row(r);
}

// The overwrite path (MXF_WRITE, aligned, N2 % VEC == 0) of the coordinate kernels as ONE-WAVE workgroups with a short argument list:
// 5 pointers + 12 scalars instead of the 25-field GramArgs, whose lazily loaded fields put ~6 dependent s_load round trips in front
// of a workgroup that lives for only 16 rows.
// Grid: x = column block (fastest: neighbouring workgroups write neighbouring 1 KiB pieces of the same rows), y = row block, z = sample
// -- no integer division in the prologue.  The diagonal term is branch-free, dadd = dscale * dadd_p[s] + jitter (the host passes
// dscale = 0 and any valid pointer when there is none), so that all scalar loads of the prologue are issued back to back.
template <typename T> struct GramLean {
    int64_t N, N2, ldk, sXs, sZs, sK, svar, sdadd, sXn, sZn;
    int tr, has_diag;
    T dscale, jitter;
};

// XF (float64 RBF only): the squared distance in the reference's own expansion form |x|^2 + |z|^2 - 2 x.z (stationary.py:98-107) --
// 1 + QT fused multiply-adds per element instead of 2 QT operations -- with the row norms from prescale_kernel, and the table-driven
// exp2 above.  In float64 the kernel is bound by its VALU work, not by HBM (~40 issue slots per 8-byte element before, ~27 with XF).
// TRC > 0 (r03): the rows per workgroup as a compile-time constant, and the diagonal term HOISTED -- a wave-uniform test in front of the row
// loop sends only the workgroups the diagonal crosses (and a ragged last row block) through the loop with the per-row check.  Same-box
// (tests/probes/gram_variants.hip "cand"): +2.0 ... 2.5 % over the run-time form at 16 rows; 12 rows measured between equal and +4 %.
template <typename T, int QT, int KIND, int XF, int TRC>
__global__ __launch_bounds__(64) void gram_lean_kernel(const T* __restrict__ Xs_all, const T* __restrict__ Zs_all, T* __restrict__ K_all,
                                                       const T* __restrict__ var, const T* __restrict__ dadd_p,
                                                       const T* __restrict__ Xn_all, const T* __restrict__ Zn_all, GramLean<T> a) {
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    const int lane = threadIdx.x;
    const int s = blockIdx.z;
    __shared__ double tab[XF ? 64 : 1];
    // (forcing the whole argument segment into SGPRs up front with an `asm volatile("" :: "s"(...))` makes the compiler fetch the x rows
    //  with per-lane vector loads instead of s_loads: 6.4 -> 5.7 TB/s, tests/probes/gram_variants.hip "force")
    const int tr = TRC > 0 ? TRC : a.tr;
    const int64_t row0 = (int64_t)blockIdx.y * tr;
    const int64_t wcol0 = (int64_t)blockIdx.x * (64 * VEC);       // first column of the wave
    const int64_t col0 = wcol0 + (int64_t)lane * VEC;
    const T variance = (KIND == MXF_K_LINEAR) ? (T)1 : var[(int64_t)s * a.svar];
    const T dadd = a.dscale * dadd_p[(int64_t)s * a.sdadd] + a.jitter;
    if constexpr (XF) {
        tab[lane] = (double)variance * mxf_exp2_neg_f64(-(double)lane * (1.0 / 64.0));       // variance 2^(lane / 64)
        __syncthreads();
    }
    if (col0 >= a.N2) return;
    T z[VEC][QT];
    {
        const T* __restrict__ Zs = Zs_all + (int64_t)s * a.sZs + col0 * QT;
#pragma unroll
        for (int i = 0; i < VEC * QT; i += VEC)
            *reinterpret_cast<V*>(&z[0][0] + i) = *reinterpret_cast<const V*>(Zs + i);
    }
    const T* __restrict__ Xrows = Xs_all + (int64_t)s * a.sXs + row0 * QT;       // wave-uniform: scalar loads
    T* __restrict__ Krow = K_all + (int64_t)s * a.sK + row0 * a.ldk + col0;
    // a diagonal term exists (decided on the host; its value may still be 0) -- TRC form: ... and crosses this workgroup's rows x columns
    const bool diag_possible = a.has_diag != 0 && (TRC == 0 || (row0 < wcol0 + 64 * VEC && row0 + tr > wcol0));
    const int rmax = (a.N - row0) < tr ? (int)(a.N - row0) : tr;
    T zz[VEC];
    const T* __restrict__ Xn = nullptr;
    if constexpr (XF == 1) {     // t = -64 (|x|^2 + |z|^2 - 2 x.z) = fma(|x|^2, -64, -64 |z|^2) + sum_q x_q (128 z_q)
        // Both norms come from prescale_kernel (for a square Gram from the SAME array) and the scalings are powers of two, so
        // K[i][j] and K[j][i] go through identical roundings: the square Gram stays bit-symmetric.
        Xn = Xn_all + (int64_t)s * a.sXn + row0;
        const T* __restrict__ Zn = Zn_all + (int64_t)s * a.sZn + col0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            zz[v] = (T)-64 * Zn[v];
#pragma unroll
            for (int q = 0; q < QT; ++q) z[v][q] *= (T)128;
        }
    }
    auto do_row = [&](int r, auto with_diag) {
        T x[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) x[q] = Xrows[r * QT + q];
        T kv[VEC];
        if constexpr (XF == 1) {
            const T xn = Xn[r];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                T acc = fma(xn, (T)-64, zz[v]);
#pragma unroll
                for (int q = 0; q < QT; ++q) acc = fma(x[q], z[v][q], acc);
                kv[v] = (T)exp2_64ths_tab((double)acc, tab);     // the table carries the variance
            }
        } else if constexpr (sizeof(T) == 4 && KIND != MXF_K_LINEAR) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int p = 0; p < VEC / 2; ++p) {
                f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const f32x2 xx = {(float)x[q], (float)x[q]};
                    const f32x2 zz = {(float)z[2 * p][q], (float)z[2 * p + 1][q]};
                    const f32x2 d = xx - zz;
                    acc2 = __builtin_elementwise_fma(d, d, acc2);
                }
                kv[2 * p] = cov_from<T, KIND>((T)acc2.x, variance);
                kv[2 * p + 1] = cov_from<T, KIND>((T)acc2.y, variance);
            }
        } else {
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
        // "+ noise / jitter on the diagonal": a scalar test (is this row inside the wave's column range?) guards the per-lane compares
        if constexpr (decltype(with_diag)::value) {
            if (diag_possible && (uint64_t)(row0 + r - wcol0) < (uint64_t)(64 * VEC)) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) if (col0 + v == row0 + r) {
                    if (XF == 1 && (a.has_diag & 2)) kv[v] = variance;     // k(x, x) = variance exactly (stationary.py:123-124), as the difference form gives
                    kv[v] += dadd;
                }
            }
        }
        V out;
        T* po = reinterpret_cast<T*>(&out);
#pragma unroll
        for (int v = 0; v < VEC; ++v) po[v] = kv[v];
        __builtin_nontemporal_store(out, reinterpret_cast<V*>(Krow + (int64_t)r * a.ldk));
    };
    if constexpr (TRC > 0) {
        if (rmax == TRC && !diag_possible) {
#pragma unroll 2
            for (int r = 0; r < TRC; ++r) do_row(r, std::false_type{});
        } else {
            for (int r = 0; r < rmax; ++r) do_row(r, std::true_type{});
        }
    } else {
#pragma unroll 2
        for (int r = 0; r < rmax; ++r) do_row(r, std::true_type{});
    }
}

// out[s][row][q] = X[s][row][q] * m_q for row < N, q < Q, else 0  (row < pad, q < QT);  m_q = c/l_q or sqrt(v_q)
// norms (optional): norms[s][row] = sum_q out[s][row][q]^2 (the QT threads of a row are neighbouring lanes)
template <typename T, int QT, int KIND>
__global__ void prescale_kernel(const T* __restrict__ X, int64_t sX, const T* __restrict__ ls, int64_t sls, int ard, int64_t N, int Q,
                                int64_t pad, T* __restrict__ out, T* __restrict__ norms = nullptr, int raw = 0,
                                const T* __restrict__ centre = nullptr, int64_t sC = 0) {
    const int s = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pad * QT) return;
    const int64_t row = i / QT;
    const int q = (int)(i % QT);
    T v = (T)0;
    if (row < N && q < Q) {
        const T l = ls[(int64_t)s * sls + (ard ? q : 0)];
        const T m = raw ? (T)1 : ((KIND == MXF_K_LINEAR) ? t_sqrt<T>(l) : coord_scale<T, KIND>() / l);      // raw: the padded copy only
        // centre (stationary kinds): one point subtracted from BOTH operands before scaling -- x / l rounds proportionally to |x| / l, so inputs
        // at an offset of 1000 units cost 4e-5 on K in float32 where centred inputs cost 1e-7 (tests/probes/offset_gram.py)
        v = (X[(int64_t)s * sX + row * Q + q] - (centre ? centre[(int64_t)s * sC + q] : (T)0)) * m;
    }
    out[(int64_t)s * pad * QT + i] = v;
    if (norms) {
        T n2 = v * v;
#pragma unroll
        for (int o = QT / 2; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
        if (q == 0) norms[(int64_t)s * pad + row] = n2;
    }
}

// generic fallback for Q > 16: one output per thread, q-loop over global memory
template <typename T, int KIND>
__global__ __launch_bounds__(256) void gram_generic_kernel(GramArgs<T> a) {
    const int MODE = a.mode;
    const int s = blockIdx.z;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= a.N2) return;
    const T* X2 = a.X2 + (int64_t)s * a.sX2 + col * a.Q;
    const T* ls = a.ls + (int64_t)s * a.sls;
    const T variance = (KIND == MXF_K_LINEAR) ? (T)1 : a.var[(int64_t)s * a.svar];
    for (int64_t row = blockIdx.y; row < a.N; row += gridDim.y) {      // grid.y is capped at 65535 row slots
        const T* X = a.X + (int64_t)s * a.sX + row * a.Q;
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
}

template <typename T, int KIND>
int launch_kind(mxf_ctx* h, GramArgs<T> a, int S, int mode, hipStream_t st) {
    constexpr int VEC = Vec16<T>::n;
    if (mode < 0 || mode > 2) MXF_FAIL(h, -2, "mxf_gram: bad mode %d", mode);
    a.mode = mode;
    if (a.Q > 16) {
        dim3 g((unsigned)((a.N2 + 255) / 256), (unsigned)(a.N < 65535 ? a.N : 65535), (unsigned)S);
        hipLaunchKernelGGL((gram_generic_kernel<T, KIND>), g, dim3(256), 0, st, a);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    a.vecst = (a.ldk % VEC == 0) && (a.sK % VEC == 0) && (((uintptr_t)a.K) % 16 == 0);
    // tuning knobs (A/B probes): rows per block, waves per block, store flavour.  Defaults measured on MI355X at N=65536, Q=8
    // (tests/probes/gram_variants.hip, tests/probes/gram_time.py): single-wave workgroups of 16 rows.
    static const int tr_env = MXF_KNOB("MXF_GRAM_TR", 0);
    static const int nw_env = MXF_KNOB("MXF_GRAM_NW", 0);
    static const int nt_env = MXF_KNOB("MXF_GRAM_NT", -1);
    // f32 RBF: 16 rows (6.05 TB/s; 32: 5.75, 64: 5.29, 8: 5.27); the VALU-heavier epilogues (Matern, all float64) prefer 64 (f64 RBF 5.25 vs 5.13)
    // (float64 RBF in the expansion form, gram_lean_kernel XF: 16 rows 5.71 ms = 6.02 TB/s, 32: 5.95, 64: 6.36 ms)
    static const int xf_tr = MXF_KNOB("MXF_GRAM_F64_EXPAND", 1);
    // (r03: float32 RBF 12 rows as a compile-time constant with the diagonal term hoisted, gram_lean_kernel TRC: between equal and +4 % against 16)
    a.tr = (tr_env == 8 || tr_env == 12 || tr_env == 16 || tr_env == 32 || tr_env == 64) ? tr_env : ((KIND == MXF_K_RBF && sizeof(T) == 4) ? 12 : (KIND == MXF_K_RBF && xf_tr) ? 16 : 64);
    a.nt = (nt_env >= 0) ? nt_env : 1;
    const int NW = (nw_env == 1 || nw_env == 4) ? nw_env : 1;
    const int64_t cw = (int64_t)NW * 64 * VEC;          // columns per workgroup
    a.ncb = (unsigned)((a.N2 + cw - 1) / cw);
    const int64_t nblk = (int64_t)a.ncb * ((a.N + a.tr - 1) / a.tr);
    if (nblk > 2147483647LL) MXF_FAIL(h, -3, "mxf_gram: problem too large for one launch");
    dim3 g((unsigned)nblk, 1, (unsigned)S);
#define GO(QT)                                                                                                        \
    do {                                                                                                              \
        const bool fast = a.vecst && mode == MXF_WRITE && a.nt && (a.N2 % VEC == 0);                                  \
        static const int lean_env = MXF_KNOB("MXF_GRAM_LEAN", 1);                     \
        static const int xf_env = MXF_KNOB("MXF_GRAM_F64_EXPAND", 1);           \
        const bool lean_ok = NW == 1 && fast && lean_env && KIND != MXF_K_BIAS && KIND != MXF_K_WHITE && (a.N + a.tr - 1) / a.tr <= 65535; \
        const bool xf = lean_ok && sizeof(T) == 8 && KIND == MXF_K_RBF && xf_env;                                     \
        T* xnorm = nullptr; T* znorm = nullptr; int64_t sxn = 0, szn = 0;                                             \
        if (KIND != MXF_K_BIAS && KIND != MXF_K_WHITE) {                                                              \
            const int64_t padr = (a.N + TR - 1) / TR * TR, padc = (a.N2 + 4 * 64 * VEC - 1) / (4 * 64 * VEC) * (4 * 64 * VEC); \
            const int Sx = (a.sX == 0 && a.sls == 0) ? 1 : S, Sz = (a.sX2 == 0 && a.sls == 0) ? 1 : S;                \
            /* the common centre of both operands: the first row of X, or of X2 when X is sampled and X2 shared (its pre-scaled copy is then \
               shared by the samples and must not depend on one of them) */                                           \
            const bool cen_x2_ = !a.square && a.sX != 0 && a.sX2 == 0 && a.X2 != nullptr;                             \
            const T* cen_ = (KIND == MXF_K_LINEAR) ? (const T*)nullptr : (cen_x2_ ? a.X2 : a.X);                      \
            const int64_t scen_ = cen_x2_ ? 0 : a.sX;                                                                 \
            const int64_t padx = a.square ? (padr > padc ? padr : padc) : padr;                                       \
            const size_t ncoord = ((size_t)Sx * padx + (a.square ? 0 : (size_t)Sz * padc)) * QT;                     \
            const size_t need = (ncoord + (xf ? (size_t)Sx * padx + (a.square ? 0 : (size_t)Sz * padc) : 0)) * sizeof(T); \
            T* buf = (T*)mxf_gram_ws(h, need);                                                                        \
            if (!buf) MXF_FAIL(h, -4, "mxf_gram: cannot allocate %zu bytes for the pre-scaled coordinates", need);    \
            xnorm = xf ? buf + ncoord : nullptr; sxn = (Sx == 1) ? 0 : padx;                                          \
            hipLaunchKernelGGL((prescale_kernel<T, QT, KIND>), dim3((unsigned)((padx * QT + 255) / 256), Sx), dim3(256), 0, st, a.X, a.sX, \
                               a.ls, a.sls, a.ard, a.N, a.Q, padx, buf, xnorm, 0, cen_, scen_);                       \
            a.Xs = buf; a.sXs = (Sx == 1) ? 0 : padx * QT;                                                            \
            if (a.square) { a.Zs = buf; a.sZs = a.sXs; znorm = xnorm; szn = sxn; }                                    \
            else {                                                                                                    \
                T* bz = buf + (size_t)Sx * padx * QT;                                                                 \
                znorm = xf ? xnorm + (size_t)Sx * padx : nullptr; szn = (Sz == 1) ? 0 : padc;                         \
                hipLaunchKernelGGL((prescale_kernel<T, QT, KIND>), dim3((unsigned)((padc * QT + 255) / 256), Sz), dim3(256), 0, st, a.X2, \
                                   a.sX2, a.ls, a.sls, a.ard, a.N2, a.Q, padc, bz, znorm, 0, cen_, scen_);            \
                a.Zs = bz; a.sZs = (Sz == 1) ? 0 : padc * QT;                                                         \
            }                                                                                                         \
        }                                                                                                             \
        if (lean_ok) {                                                                                                \
            GramLean<T> l;                                                                                            \
            l.N = a.N; l.N2 = a.N2; l.ldk = a.ldk; l.sXs = a.sXs; l.sZs = a.sZs; l.sK = a.sK; l.svar = a.svar; l.tr = a.tr; \
            const bool hd = a.square && a.dadd != nullptr;                                                            \
            l.sdadd = hd ? a.sdadd : 0; l.dscale = hd ? (T)1 : (T)0; l.jitter = a.square ? a.jitter : (T)0;           \
            l.has_diag = ((hd || l.jitter != (T)0) ? 1 : 0) | ((xf && a.square) ? 2 : 0); l.sXn = sxn; l.sZn = szn;   \
            const T* dptr = hd ? a.dadd : (a.var ? a.var : a.Xs);          /* any readable word when there is no diagonal term */ \
            dim3 gl(a.ncb, (unsigned)((a.N + a.tr - 1) / a.tr), (unsigned)S);                                         \
            if constexpr (sizeof(T) == 8 && KIND == MXF_K_RBF) {                                                      \
                if (xf && a.tr == 16) hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 1, 16>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)xnorm, (const T*)znorm, l); \
                else if (xf) hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 1, 0>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)xnorm, (const T*)znorm, l); \
                else hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 0, 0>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)nullptr, (const T*)nullptr, l); \
            } else if constexpr (sizeof(T) == 4 && KIND == MXF_K_RBF) {                                               \
                if (a.tr == 12) hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 0, 12>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)nullptr, (const T*)nullptr, l); \
                else if (a.tr == 16) hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 0, 16>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)nullptr, (const T*)nullptr, l); \
                else hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 0, 0>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)nullptr, (const T*)nullptr, l); \
            } else hipLaunchKernelGGL((gram_lean_kernel<T, QT, KIND, 0, 0>), gl, dim3(64), 0, st, a.Xs, a.Zs, a.K, a.var, dptr, (const T*)nullptr, (const T*)nullptr, l); \
        } else if (NW == 1 && fast) hipLaunchKernelGGL((gram_kernel<T, QT, KIND, 1, true>), g, dim3(64), 0, st, a, a.Xs, a.Zs, a.K); \
        else if (NW == 1) hipLaunchKernelGGL((gram_kernel<T, QT, KIND, 1, false>), g, dim3(64), 0, st, a, a.Xs, a.Zs, a.K);            \
        else if (fast) hipLaunchKernelGGL((gram_kernel<T, QT, KIND, 4, true>), g, dim3(256), 0, st, a, a.Xs, a.Zs, a.K);               \
        else hipLaunchKernelGGL((gram_kernel<T, QT, KIND, 4, false>), g, dim3(256), 0, st, a, a.Xs, a.Zs, a.K);                        \
    } while (0)
    if (KIND == MXF_K_BIAS || KIND == MXF_K_WHITE) GO(2);
    else if (a.Q <= 2) GO(2);
    else if (a.Q <= 4) GO(4);
    else if (a.Q <= 8) GO(8);
    else GO(16);
#undef GO
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// ---- Gram matrix written as split planes (the operand formats of gemm_split.hip: NP = 3 bf16 terms, NP = 2 scaled f16 terms) ----------
// operand element (r, k) = cov(xmin[r], xmaj[k]); plane p element (r, k) at ((k / 16) * R + r) * 16 + k % 16.
// Thread <-> (minor index r, k half): per 16-wide k block a thread evaluates 8 covariances, splits each f32 value exactly into
// h + m + l (bf16 each) and writes ONE 16-byte unit per plane; a wave writes 1 KB contiguous per plane per k block.
// HBM-write bound: 2 NP bytes per element (8.6 GB at M = 1024 x 2.1 M columns for NP = 2).
typedef unsigned int gp_u32x4 __attribute__((ext_vector_type(4)));
// NP = 2 (f16x2 format of gemm_split.hip): the planes hold cov / variance * 2^14 as hi + lo (f16 each); the consumer multiplies by
// variance * 2^-14 (unit-variance covariances are <= 1, so the format's power-of-two scale is known without a reduction).
// PT > 0: the same pass also forms  U[p][r] = sum_k w[k][p] cov(xmin[r], xmaj[k])  (the row w^T Kuf of the SVGP step, svgp_regression.py:98:
// Kuf^T Kuu^-1 mu) from the f32 covariances it has in registers -- the separate 8.6 GB read of the planes that product used to cost
// (1.45 ms at the bench size) is gone.  Needs grid.y == 1 (every block walks all k blocks of its rows).
template <int QT, int KIND, int NP, int PT>
__global__ __launch_bounds__(256) void gram_planes_kernel(int64_t R, int64_t Kn, const float* __restrict__ Xmin_s, const float* __restrict__ Xmaj_s,
                                                          const float* __restrict__ var, unsigned short* __restrict__ P, int64_t pstride,
                                                          int chunks_per_block, const float* __restrict__ wk, int Pw, float* __restrict__ U,
                                                          int64_t ldU) {
    constexpr int CH = 16;                                   // k blocks per staged chunk (256 major points)
    __shared__ __attribute__((aligned(16))) float xs[CH * 16 * QT];
    __shared__ float ws[PT > 0 ? CH * 16 * PT : 1];          // w of the staged chunk, [k][p]
    float uacc[PT > 0 ? PT : 1];
#pragma unroll
    for (int p = 0; p < (PT > 0 ? PT : 1); ++p) uacc[p] = 0.f;
    const int tid = threadIdx.x, rl = tid >> 1, half = tid & 1;
    const int64_t r = (int64_t)blockIdx.x * 128 + rl;
    const bool rvalid = r < R;
    const float variance = NP == 2 ? 16384.f : var[0];
    float z[QT];
#pragma unroll
    for (int q = 0; q < QT; q += 4) *reinterpret_cast<f32x4_t*>(&z[q]) = *reinterpret_cast<const f32x4_t*>(Xmin_s + r * QT + q);   // padded
    const int64_t K16 = (Kn + 15) / 16;
    for (int c = 0; c < chunks_per_block; ++c) {
        const int64_t kb0 = ((int64_t)blockIdx.y * chunks_per_block + c) * CH;
        if (kb0 >= K16) break;
        __syncthreads();
        for (int i = tid * 4; i < CH * 16 * QT; i += 256 * 4)
            *reinterpret_cast<f32x4_t*>(&xs[i]) = *reinterpret_cast<const f32x4_t*>(Xmaj_s + kb0 * 16 * QT + i);                   // padded
        if (PT > 0) {
            for (int i = tid; i < CH * 16 * PT; i += 256) {
                const int64_t k = kb0 * 16 + i / PT;
                const int p = i % PT;
                ws[i] = (k < Kn && p < Pw) ? wk[k * Pw + p] : 0.f;
            }
        }
        __syncthreads();
        const int nkb = (int)((K16 - kb0) < CH ? (K16 - kb0) : CH);
        // (FULL: every major point of the chunk exists -- always, when Kn is a multiple of 16 -- so the per-element range test, a 64-bit
        //  compare + select per covariance, 8 % of the kernel's instructions, is compiled out)
        auto chunk_body = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        for (int kbl = 0; kbl < nkb; ++kbl) {
            gp_u32x4 uh, um, ul;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                float kv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int nl = kbl * 16 + half * 8 + j + e;
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < QT; q += 2) {
                        const f32x2 xx = {xs[nl * QT + q], xs[nl * QT + q + 1]};
                        const f32x2 zz = {z[q], z[q + 1]};
                        const f32x2 d = xx - zz;
                        acc2 = __builtin_elementwise_fma(d, d, acc2);
                    }
                    const float red = acc2.x + acc2.y;
                    kv[e] = (FULL || kb0 * 16 + nl < Kn) ? cov_from<float, KIND>(red, variance) : 0.f;
                    if (PT > 0) {
#pragma unroll
                        for (int p = 0; p < PT; ++p) uacc[p] = fmaf(ws[nl * PT + p], kv[e], uacc[p]);
                    }
                }
                unsigned hh = 0, mm = 0, ll = 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (NP == 2) {
                        const _Float16 fh = (_Float16)kv[e];
                        const _Float16 fl = (_Float16)(kv[e] - (float)fh);
                        hh |= (unsigned)__builtin_bit_cast(unsigned short, fh) << (16 * e);
                        mm |= (unsigned)__builtin_bit_cast(unsigned short, fl) << (16 * e);
                        continue;
                    }
                    const __bf16 bh = (__bf16)kv[e];
                    const float r1 = kv[e] - (float)bh;
                    const __bf16 bm = (__bf16)r1;
                    const float r2 = r1 - (float)bm;
                    const __bf16 bl = (__bf16)r2;
                    hh |= (unsigned)__builtin_bit_cast(unsigned short, bh) << (16 * e);
                    mm |= (unsigned)__builtin_bit_cast(unsigned short, bm) << (16 * e);
                    ll |= (unsigned)__builtin_bit_cast(unsigned short, bl) << (16 * e);
                }
                uh[j / 2] = hh; um[j / 2] = mm; ul[j / 2] = ll;
            }
            if (rvalid) {
                unsigned short* dst = P + ((kb0 + kbl) * R + r) * 16 + half * 8;
                __builtin_nontemporal_store(uh, reinterpret_cast<gp_u32x4*>(dst));
                __builtin_nontemporal_store(um, reinterpret_cast<gp_u32x4*>(dst + pstride));
                if (NP == 3) __builtin_nontemporal_store(ul, reinterpret_cast<gp_u32x4*>(dst + 2 * pstride));
            }
        }
        };
        if ((kb0 + nkb) * 16 <= Kn) chunk_body(std::true_type{}); else chunk_body(std::false_type{});
    }
    if (PT > 0) {      // the two threads of a row hold the two k halves: fold them, undo the plane scaling, one store per (row, p)
        const float sc = NP == 2 ? var[0] * (1.f / 16384.f) : 1.f;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const float v = uacc[p] + __shfl_xor(uacc[p], 1, 64);
            if (half == 0 && rvalid && p < Pw) U[(int64_t)p * ldU + r] = v * sc;
        }
    }
}

// (r03) NP = 2 as ONE-WAVE workgroups, lane <-> minor index r (64 rows per wave), the sixteen major points of a k block wave-uniform
// (scalar loads, no LDS staging, no barrier -- the layout of gram_lean_kernel).  A lane evaluates the 16 covariances of its row, i.e. BOTH
// 16-byte halves of its 32-byte piece; v_permlane32_swap then trades half 1 of the rows of lanes 0..31 for half 0 of the rows of lanes
// 32..63, so each of the two store instructions per plane covers 1 KB CONTIGUOUS (32 rows x 32 bytes).  (Without the swap each store
// covers 16-byte pieces at a 32-byte stride: measured r02, 30 -> 40 ms per step.)  hi + lo come from the packed conversions
// (v_cvt_pk_f16_f32: a plane's dword directly); the fused row U sums its sixteen terms per k block in index order.
// Measured (tests/probes/planes_lean.sh, planes_store.hip): 1.82 / 1.69 ms for the two 8.6 GB passes of the bench step, the same as the staged
// kernel -- neither the LDS staging nor the VALU count (16 -> 14 instructions per covariance here) is what bounds them; the store-only twin
// of the same pattern takes 1.45 - 1.65 ms alone, memset 1.36 ms; PLAIN instead of non-temporal stores: 2.1 - 2.4 ms inside the step.
// ACC (r04): the squared distance as sum_q ((x_q - z_q)^2) (c / l_q)^2 from the RAW coordinates -- difference first, scale after -- instead of
// the difference of pre-scaled coordinates: the rounding of x / l (2^-24 |x / l|) no longer enters r^2 of NEAR pairs, the ones that carry the
// covariance.  At length-scale 0.2 on [-2, 2] (a deep GP's hidden layer) the pre-scaled form's Gram error, amplified by |T| ~ sqrt(cond) in
// q_n = k_n . T_n, was 2/3 of the whitened tier's ELBO error (3.9e-5 -> 4e-6 relative; tests/test_gpu_f32_guard.py two-layer case).
// One more packed multiply per two coordinates.
// MXF_PLANES_WAVES (compile time, r05 experiment): at most this many waves of the planes pass per SIMD.  Its one-wave workgroups otherwise take
// every wave slot (and with ~64 registers each the whole register file) of every CU, and the float64 workgroups of the Kuu chain that runs
// next to it (4 waves of ~128 registers, tens of KB of LDS) wait for a CU to drain by chance -- the r05 timelines show a 60 us GEMM of
// that chain taking 1.26 ms next to this pass.
#ifndef MXF_PLANES_WAVES
#define MXF_PLANES_WAVES 0
#endif
#if MXF_PLANES_WAVES > 0
#define MXF_PLANES_OCC __attribute__((amdgpu_waves_per_eu(MXF_PLANES_WAVES, MXF_PLANES_WAVES)))
#else
#define MXF_PLANES_OCC
#endif
template <int QT, int KIND, int PT, bool ACC = false, bool PERS = false>
__global__ __launch_bounds__(64) MXF_PLANES_OCC void gram_planes_lean_kernel(int64_t R, int64_t Kn, const float* __restrict__ Xmin_s, const float* __restrict__ Xmaj_s,
                                                              const float* __restrict__ var, unsigned short* __restrict__ P, int64_t pstride,
                                                              int kb_per_block, const float* __restrict__ wk, int Pw, float* __restrict__ U,
                                                              int64_t ldU, const float* __restrict__ ls = nullptr, int ard = 0, int Q = 0,
                                                              const float* __restrict__ majs = nullptr, const float* __restrict__ mins = nullptr,
                                                              int64_t period = 1, int64_t pnbx = 0, int pnby = 0) {
    // pnbx > 0 (r06, PT == 0 only): PERSISTENT form -- gridDim.x waves walk the pnbx x pnby (row block, k chunk) items, row block fastest.  The
    // launcher sizes the grid to fewer waves than the chip holds, so that the float64 workgroups of the Kuu chain that runs next to this pass
    // (four waves + LDS each) find room on every CU instead of waiting for four wave slots of one CU to drain at the same moment.
    // majs / mins (ACC form only; the streaming heteroscedastic SVGP bound, svgp_regression.py:61-67): every covariance is multiplied by
    // majs[major index % period] (wave-uniform) and / or mins[minor index % period] (per lane) before it is split into planes -- and before
    // it enters the fused row U -- i.e. the planes hold K diag(s) resp. diag(s) K for per-row weights s <= 1
    const int lane = threadIdx.x;
    float s2[QT];
    if constexpr (ACC) {
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float m = q < Q ? coord_scale<float, KIND>() / ls[ard ? q : 0] : 0.f;      // wave-uniform: scalar loads
            s2[q] = m * m;
        }
    }
    constexpr bool pers = PERS && PT == 0;      // (a compile-time form: the loop around the body cost the plain instance 7 % -- 1.72 -> 1.85 ms in the step)
    int64_t item = blockIdx.x;
    do {
    const int64_t bxi = pers ? item % pnbx : (int64_t)blockIdx.x;
    const int byi = pers ? (int)(item / pnbx) : (int)blockIdx.y;
    const int64_t r0 = bxi * 64, r = r0 + lane;
    float z[QT];
#pragma unroll
    for (int q = 0; q < QT; q += 4) *reinterpret_cast<f32x4_t*>(&z[q]) = *reinterpret_cast<const f32x4_t*>(Xmin_s + r * QT + q);   // padded
    float mscale = 1.f;
    if constexpr (ACC) { if (mins) mscale = mins[(r < R ? r : R - 1) % period]; }
    float uacc[PT > 0 ? PT : 1];
#pragma unroll
    for (int p = 0; p < (PT > 0 ? PT : 1); ++p) uacc[p] = 0.f;
    const int64_t K16 = (Kn + 15) / 16;
    const int64_t kb_begin = (int64_t)byi * kb_per_block;
    const int64_t kb_end = (kb_begin + kb_per_block) < K16 ? (kb_begin + kb_per_block) : K16;
    // store targets after the swap: instruction 1 <-> row r0 + lane % 32, instruction 2 <-> 32 rows further; the half is lane / 32
    const int64_t ra = r0 + (lane & 31), rb = ra + 32;
    const int hsel = lane >> 5;
    auto kb_body = [&](int64_t kb, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const float* __restrict__ xk = Xmaj_s + kb * 16 * QT;                    // wave-uniform: scalar loads
        unsigned hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            f32x2 kv2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int nl = 2 * j + e;
                f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < QT; q += 2) {
                    const f32x2 xx = {xk[nl * QT + q], xk[nl * QT + q + 1]};
                    const f32x2 zz = {z[q], z[q + 1]};
                    const f32x2 d = xx - zz;
                    if constexpr (ACC) { const f32x2 ss = {s2[q], s2[q + 1]}; acc2 = __builtin_elementwise_fma(d * d, ss, acc2); }
                    else acc2 = __builtin_elementwise_fma(d, d, acc2);
                }
                const float red = acc2.x + acc2.y;
                float kv = (FULL || kb * 16 + nl < Kn) ? cov_from<float, KIND>(red, 16384.f) : 0.f;
                if constexpr (ACC) {
                    if (majs) { const int64_t km = FULL ? kb * 16 + nl : ((kb * 16 + nl < Kn) ? kb * 16 + nl : Kn - 1); kv *= majs[km % period]; }
                    if (mins) kv *= mscale;
                }
                if (PT > 0) {
                    const int64_t kk = FULL ? kb * 16 + nl : ((kb * 16 + nl < Kn) ? kb * 16 + nl : Kn - 1);     // (kv == 0 beyond Kn)
#pragma unroll
                    for (int p = 0; p < PT; ++p) {
                        const float wv = (PT == 1 || p < Pw) ? wk[kk * Pw + (PT == 1 ? 0 : p)] : 0.f;       // wave-uniform
                        uacc[p] = fmaf(wv, kv, uacc[p]);
                    }
                }
                kv2[e] = kv;
            }
            // hi + lo two values at a time: the packed conversions (v_cvt_pk_f16_f32) deliver the dword of the plane directly
            const f16x2 fh = __builtin_convertvector(kv2, f16x2);
            const f16x2 fl = __builtin_convertvector(kv2 - __builtin_convertvector(fh, f32x2), f16x2);
            const unsigned hh = __builtin_bit_cast(unsigned, fh), ll = __builtin_bit_cast(unsigned, fl);
            hi[j] = hh; lo[j] = ll;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hi[i]), "+v"(hi[4 + i]));
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo[i]), "+v"(lo[4 + i]));
        }
        unsigned short* base = P + (kb * R) * 16 + hsel * 8;
        if (ra < R) {
            const gp_u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
            __builtin_nontemporal_store(vh, reinterpret_cast<gp_u32x4*>(base + ra * 16));
            __builtin_nontemporal_store(vl, reinterpret_cast<gp_u32x4*>(base + ra * 16 + pstride));
        }
        if (rb < R) {
            const gp_u32x4 vh = {hi[4], hi[5], hi[6], hi[7]}, vl = {lo[4], lo[5], lo[6], lo[7]};
            __builtin_nontemporal_store(vh, reinterpret_cast<gp_u32x4*>(base + rb * 16));
            __builtin_nontemporal_store(vl, reinterpret_cast<gp_u32x4*>(base + rb * 16 + pstride));
        }
    };
    const int64_t kb_full = Kn / 16 < kb_end ? Kn / 16 : kb_end;           // k blocks whose sixteen major points all exist
    int64_t kb = kb_begin;
    for (; kb < kb_full; ++kb) kb_body(kb, std::true_type{});
    for (; kb < kb_end; ++kb) kb_body(kb, std::false_type{});
    if (PT > 0) {
        const float sc = var[0] * (1.f / 16384.f);
        if (r < R) {
#pragma unroll
            for (int p = 0; p < PT; ++p) if (p < Pw) U[(int64_t)p * ldU + r] = uacc[p] * sc;
        }
    }
    item += gridDim.x;
    } while (pers && item < pnbx * (int64_t)pnby);
}

template <int KIND>
int gram_planes_kind(mxf_ctx* h, int64_t R, int64_t Kn, int Q, const float* Xmin, const float* Xmaj, const float* ls, int ard,
                     const float* var, unsigned short* planes, int64_t pstride, float* scratch, hipStream_t st, int mode,
                     const float* wk, int Pw, float* U, int64_t ldU, const float* majs, const float* mins, int64_t period) {
    const int QT = Q <= 8 ? 8 : 16;
    const int64_t padr = (R + 127) / 128 * 128, padk = ((Kn + 15) / 16 + 15) / 16 * 256;
    float* buf = scratch;     // (padr + padk) * QT floats, caller-owned: two of these run concurrently on different streams
    float* bmaj = buf + (size_t)padr * QT;
    const int64_t K16 = (Kn + 15) / 16, chunks = (K16 + 15) / 16, rblocks = padr / 128;
    int cpb = 1;
    while (rblocks * ((chunks + cpb - 1) / cpb) > 16384 && cpb < chunks) cpb *= 2;      // fewer, longer blocks once the chip is full
    const bool fuse_u = U != nullptr;
    if (fuse_u) {
        if (Pw > 8 || !wk) MXF_FAIL(h, -3, "gram planes: fused w^T K needs w and P <= 8");
        cpb = (int)chunks;                                                              // every block walks all k blocks of its rows
    }
    dim3 grid((unsigned)rblocks, (unsigned)((chunks + cpb - 1) / cpb));
    if (grid.y > 65535u) MXF_FAIL(h, -3, "gram planes: grid too large");
    // (measured and dropped: one-wave workgroups with lane <-> row and the k-side points through scalar loads, as in the Gram kernel -- each
    //  store instruction then covers 16-byte pieces at a 32-byte stride and the step went from 30.1 to 40.0 ms; the (row, k half) <-> thread
    //  mapping below writes whole lines per instruction)
    // r03: the one-wave form for the f16x2 planes (probe builds: MXF_PLANES_LEAN=0 selects the staged kernel)
    const bool lean = mode == MXF_SPLIT_F16X2 && MXF_KNOB("MXF_PLANES_LEAN", 1) != 0;
    const int raw = (lean && MXF_KNOB("MXF_PLANES_ACC", 1) != 0) ? 1 : 0;       // difference-then-scale distances (gram_planes_lean_kernel ACC)
    if ((majs || mins) && !raw) MXF_FAIL(h, -3, "gram planes: per-row weights need the f16x2 lean kernel");
    int kbpb = fuse_u ? (int)K16 : (int)MXF_KNOB("MXF_PLANES_KB", 8);
    while (!fuse_u && (K16 + kbpb - 1) / kbpb > 65535) kbpb *= 2;
    dim3 lgrid((unsigned)((R + 63) / 64), (unsigned)((K16 + kbpb - 1) / kbpb));
    // MXF_PLANES_PERSIST = w > 0 (r06, probe builds): the pass as w persistent waves per CU (gram_planes_lean_kernel pnbx) once it has more items than
    // that.  Measured and NOT kept (same box, alternating, tests/probes/r06_planes_persist.sh; parity tests pass with it): whitened 32-sample step
    // 31.4-31.6 ms without, 31.3-31.7 with 28 waves per CU, 31.9 with 16, 34 with 24; the explicit 32-sample step 22.5-22.9 -> 24.5-24.8 (28) --
    // the chain next to the pass does not get faster and the pass itself loses its dispatch-order store pattern.
    static const int pers_env = (int)MXF_KNOB("MXF_PLANES_PERSIST", 0);
    int64_t pers_grid = 0;
    if (pers_env > 0 && !fuse_u && (int64_t)lgrid.x * lgrid.y > (int64_t)256 * pers_env * 2) pers_grid = (int64_t)256 * pers_env;
#define GO(QTV)                                                                                                                       \
    do {                                                                                                                              \
        hipLaunchKernelGGL((prescale_kernel<float, QTV, KIND>), dim3((unsigned)((padr * QTV + 255) / 256), 1), dim3(256), 0, st, Xmin, (int64_t)0, ls, \
                           (int64_t)0, ard, R, Q, padr, buf, (float*)nullptr, raw);                                                   \
        hipLaunchKernelGGL((prescale_kernel<float, QTV, KIND>), dim3((unsigned)((padk * QTV + 255) / 256), 1), dim3(256), 0, st, Xmaj, (int64_t)0, ls, \
                           (int64_t)0, ard, Kn, Q, padk, bmaj, (float*)nullptr, raw);                                                 \
        if (lean && raw && fuse_u && Pw == 1)                                                                                         \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 1, true>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU, ls, ard, Q, majs, mins, period); \
        else if (lean && raw && fuse_u)                                                                                               \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 8, true>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU, ls, ard, Q, majs, mins, period); \
        else if (lean && raw && pers_grid > 0)                                                                                        \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 0, true, true>), dim3((unsigned)pers_grid), dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU, ls, ard, Q, majs, mins, period, (int64_t)lgrid.x, (int)lgrid.y); \
        else if (lean && raw)                                                                                                         \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 0, true>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU, ls, ard, Q, majs, mins, period); \
        else if (lean && fuse_u && Pw == 1)                                                                                           \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 1>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU); \
        else if (lean && fuse_u)                                                                                                      \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 8>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU); \
        else if (lean)                                                                                                                \
            hipLaunchKernelGGL((gram_planes_lean_kernel<QTV, KIND, 0>), lgrid, dim3(64), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, kbpb, wk, Pw, U, ldU); \
        else if (mode == MXF_SPLIT_F16X2 && fuse_u && Pw == 1)                                                                             \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 2, 1>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
        else if (mode == MXF_SPLIT_F16X2 && fuse_u)                                                                                   \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 2, 8>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
        else if (mode == MXF_SPLIT_F16X2)                                                                                             \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 2, 0>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
        else if (fuse_u && Pw == 1)                                                                                                   \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 3, 1>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
        else if (fuse_u)                                                                                                              \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 3, 8>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
        else                                                                                                                          \
            hipLaunchKernelGGL((gram_planes_kernel<QTV, KIND, 3, 0>), grid, dim3(256), 0, st, R, Kn, (const float*)buf, (const float*)bmaj, var, planes, pstride, cpb, wk, Pw, U, ldU); \
    } while (0)
    if (QT == 8) GO(8); else GO(16);
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

// Gram matrix cov(xmin[r], xmaj[k]) (r < R, k < Kn) as split planes in either format (float32 inputs, stationary kernels); see
// gram_planes_kernel.  planes: 3 * pstride elements, pstride = mxf_split_plane_elems(R, Kn).
size_t mxf_gram_planes_scratch_bytes(int64_t R, int64_t Kn, int Q) {
    const int QT = Q <= 8 ? 8 : 16;
    const int64_t padr = (R + 127) / 128 * 128, padk = ((Kn + 15) / 16 + 15) / 16 * 256;
    return (size_t)(padr + padk) * QT * sizeof(float);
}

int mxf_gram_planes_internal(mxf_ctx* h, int kind, int64_t R, int64_t Kn, int Q, const float* Xmin, const float* Xmaj, const float* ls,
                             int ard, const float* var, unsigned short* planes, int64_t pstride, float* scratch, hipStream_t st, int mode,
                             const float* wk, int Pw, float* U, int64_t ldU, const float* majs, const float* mins, int64_t period) {
    if (R <= 0 || Kn <= 0) return 0;
    if (!scratch) MXF_FAIL(h, -2, "gram planes: scratch of mxf_gram_planes_scratch_bytes() bytes required");
    if (Q > 16) MXF_FAIL(h, -3, "gram planes: Q > 16 not supported");
    switch (kind) {
        case MXF_K_RBF: return gram_planes_kind<MXF_K_RBF>(h, R, Kn, Q, Xmin, Xmaj, ls, ard, var, planes, pstride, scratch, st, mode, wk, Pw, U, ldU, majs, mins, period);
        case MXF_K_MATERN12: return gram_planes_kind<MXF_K_MATERN12>(h, R, Kn, Q, Xmin, Xmaj, ls, ard, var, planes, pstride, scratch, st, mode, wk, Pw, U, ldU, majs, mins, period);
        case MXF_K_MATERN32: return gram_planes_kind<MXF_K_MATERN32>(h, R, Kn, Q, Xmin, Xmaj, ls, ard, var, planes, pstride, scratch, st, mode, wk, Pw, U, ldU, majs, mins, period);
        case MXF_K_MATERN52: return gram_planes_kind<MXF_K_MATERN52>(h, R, Kn, Q, Xmin, Xmaj, ls, ard, var, planes, pstride, scratch, st, mode, wk, Pw, U, ldU, majs, mins, period);
    }
    MXF_FAIL(h, -2, "gram planes: stationary kernels only (kind %d)", kind);
}

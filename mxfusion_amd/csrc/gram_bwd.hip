// Reverse mode of the stationary Gram build (gfx950), plain and SVGP-fused.
//
// Plain:  dK -> dX, dX2, dlengthscale, dvariance in ONE streaming pass over dK (HBM-read bound:
//         S*N*N2*sizeof(T) bytes), recomputing k(x,z) on the fly.  This is what MXNet autograd does through
//         stationary.py:92-106 + rbf.py:71-72 / matern.py:84-151 with ~10 materialised N^2 temporaries.
// Fused (SVGP data term, svgp_regression.py:85-107 reverse mode): dKuf is never materialised; it is formed
//         per element from T = H0*Kuf as  a1*beta*(P*T[m,n] + w[m,:].e[n,:]),  and the same pass accumulates
//         q_n = k_n^T H0 k_n, |e_n|^2, R = Kuf E, dY.
//
// Mapping (wave64): lane <-> column (coalesced row reads of dK / T); a block owns CT column tiles of 256
// columns and a band of RB rows.  Column-side sums (dX2) live in VGPRs and are stored once.  Row-side sums
// (dX / dZ, R) are wavefront-shuffle reduced per row and accumulated in an LDS band racc[RB][.] across all
// the block's column tiles, so global atomics are RB*(Q+P) per block instead of per wave-row.
#include "common.h"
#include "internal.h"
#include <stdlib.h>

namespace {

constexpr int TRB = 64;
constexpr int PMAX_ALL = 8;

template <typename T>
struct GramBwdArgs {
    const T* X; const T* X2; const T* ls; const T* var; const T* dK;
    T* dX; T* dX2; T* dls; T* dvar;
    int64_t N, N2, lddk;
    int64_t sX, sX2, sls, svar, sdK;
    int Q, ard, square;
    int64_t RB; int CT;
    // fused SVGP extras
    const T* U; const T* Y; const T* w; const T* noise;
    T* dY; T* R; double* scal;
    int64_t sY, B;
    double a1;
    int P, dY_shared;
    int tblk;        // fused: T (= dK) in 16-column blocks, element (m, n) at ((n / 16) * N + m) * 16 + n % 16 (the split GEMM's blocked output)
    int sym;         // r06, square case: the caller vouches that dK is symmetric -- the row-side sum of row i then equals the column-side sum of
                     // column i, so the row side (a reduce-scatter and an LDS atomic per row and wave: what bounds this kernel) is skipped and
                     // the column side counts twice (exact GP N = 8192 float64: 0.59 -> 0.51 ms).
                     // sym == 2: only the LOWER triangle of dK is valid (the upper one may hold anything): pairs (row, col < row) count twice, the
                     // diagonal once, the rest not at all -- half the pairs, both sides; column tiles above a block's row band are skipped (tiled kernel, Q <= 16 only)
};

__device__ __forceinline__ void lds_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add(double* p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// e^x for x <= 0 as the hardware exp2 of x log2(e) (1 ulp of exp2 + the rounding of the product: relative error ~|x| 2^-24, the same
// formulation -- and error level -- as the forward Gram kernels' exp2 of pre-scaled coordinates; the library exp costs ~15 instructions)
__device__ __forceinline__ float expnp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double expnp(double x) { return mxf_exp_nonpos_f64(x); }      // arguments here are never positive

// unit-variance covariance k and slope dk/d(r2) (r2 in lengthscale-scaled coordinates)
template <typename T, int KIND>
__device__ __forceinline__ void cov_and_slope(T r2, T& k, T& w) {
    if (KIND == MXF_K_RBF) { k = expnp((T)-0.5 * r2); w = (T)-0.5 * k; return; }
    const bool clipped = r2 < (T)1e-14;
    const T r = sqrt(clipped ? (T)1e-14 : r2);
    if (KIND == MXF_K_MATERN12) { k = expnp(-r); w = clipped ? (T)0 : -k / ((T)2 * r); return; }
    if (KIND == MXF_K_MATERN32) {
        const T s3 = (T)1.7320508075688772, e = expnp(-s3 * r);
        k = ((T)1 + s3 * r) * e; w = clipped ? (T)0 : (T)-1.5 * e; return;
    }
    const T s5 = (T)2.23606797749979, e = expnp(-s5 * r);   // MATERN52 (matern.py:85-87: un-clipped r2 in the 5/3 term)
    k = ((T)1 + s5 * r + (T)(5.0 / 3.0) * r2) * e;
    w = clipped ? (T)(5.0 / 3.0) * e : (T)(-5.0 / 6.0) * ((T)1 + s5 * r) * e;
}

template <typename T, int QT, int KIND, int PT>   // PT = 0: plain; PT > 0: SVGP-fused with P <= PT outputs
__global__ __launch_bounds__(256) void gram_bwd_kernel(GramBwdArgs<T> a) {
    constexpr bool FUSED = PT > 0;
    constexpr int PMAX = FUSED ? PT : 1;
    constexpr int QA = QT + (FUSED ? PMAX : 0);
    constexpr bool PACKED = sizeof(T) == 4 && QT >= 2;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* racc = reinterpret_cast<T*>(smem_raw);           // [RB][QA]
    T* xs = racc + a.RB * QA;                           // [TRB][QT]
    T* wsm = xs + TRB * QT;                             // [TRB][PMAX] (fused)
    T* red = wsm + (FUSED ? TRB * PMAX : 0);            // [16]
    double* redd = reinterpret_cast<double*>(red + 16); // [16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int s = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.y * a.RB;
    const int64_t rend = (r0 + a.RB < a.N) ? r0 + a.RB : a.N;
    const int Q = a.Q, P = FUSED ? a.P : 0;
    const T* __restrict__ X = a.X + (int64_t)s * a.sX;
    const T* __restrict__ X2 = a.X2 + (int64_t)s * a.sX2;
    const T* __restrict__ ls = a.ls + (int64_t)s * a.sls;
    const T variance = a.var[(int64_t)s * a.svar];
    const T* __restrict__ dK = a.dK + (int64_t)s * a.sdK;

    T il[QT], cen[QT];
    // stationary kinds: both operands are centred on the first row of X before they are scaled -- distances do not change, but x / l carries
    // a rounding error proportional to |x| / l (inputs at an offset of 1000 units: 4e-5 on K and its gradients in float32 instead of 1e-7)
#pragma unroll
    for (int q = 0; q < QT; ++q) { il[q] = (q < Q) ? (T)1 / ls[a.ard ? q : 0] : (T)0; cen[q] = (KIND != MXF_K_LINEAR && q < Q) ? X[q] : (T)0; }
    for (int64_t i = tid; i < (rend - r0) * QA; i += 256) racc[i] = (T)0;

    T gl[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) gl[q] = 0;
    T gvar = 0;
    const T beta = FUSED ? (T)1 / a.noise[0] : (T)0;
    const T c1 = FUSED ? (T)a.a1 * beta : (T)0;
    const bool rowside = a.dX != nullptr && !(a.square && a.sym == 1);
    const bool lower = a.square && a.sym == 2;

    for (int ct = 0; ct < a.CT; ++ct) {
        const int64_t tile0 = ((int64_t)blockIdx.x * a.CT + ct) * 256;
        if (tile0 >= a.N2) break;
        if (lower && tile0 >= rend) break;                 // the rest of this block's column tiles lie above its row band
        const int64_t col = tile0 + tid;
        const bool cvalid = col < a.N2;
        T z[QT], gz[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) { z[q] = (cvalid && q < Q) ? (X2[col * Q + q] - cen[q]) * il[q] : (T)0; gz[q] = 0; }
        T e[PMAX];
        double e2 = 0;
        T qn = 0;
        if (FUSED) {
            const int64_t sm = cvalid ? col / a.B : 0, nb = cvalid ? col % a.B : 0;
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                e[p] = 0;
                if (p < P && cvalid) {
                    e[p] = a.Y[sm * a.sY + nb * P + p] - a.U[(int64_t)p * a.lddk + col];
                    e2 += (double)e[p] * (double)e[p];
                    if (a.dY && blockIdx.y == 0) {      // one row band owns the per-column outputs
                        const T g = -c1 * e[p];
                        if (a.dY_shared) atomic_add(a.dY + nb * P + p, g); else a.dY[col * P + p] = g;
                    }
                }
            }
        }
        for (int64_t rt = r0; rt < rend; rt += TRB) {
            __syncthreads();
            for (int i = tid; i < TRB * QT; i += 256) {
                const int r = i / QT, q = i % QT;
                const int64_t row = rt + r;
                xs[i] = (row < rend && q < Q) ? (X[row * Q + q] - (KIND != MXF_K_LINEAR ? X[q] : (T)0)) / ls[a.ard ? q : 0] : (T)0;
            }
            if (FUSED) {
                for (int i = tid; i < TRB * PMAX; i += 256) {
                    const int r = i / PMAX, p = i % PMAX;
                    const int64_t row = rt + r;
                    wsm[i] = (row < rend && p < P) ? a.w[row * P + p] : (T)0;
                }
            }
            __syncthreads();
            const int rmax = (int)((rend - rt) < TRB ? (rend - rt) : TRB);
            constexpr int UNR = 4;                 // rows of dK / T fetched ahead of their use (hides the HBM latency of the row loads)
            for (int r0_ = 0; r0_ < rmax; r0_ += UNR) {
              T pre[UNR];
#pragma unroll
              for (int u = 0; u < UNR; ++u)
                  pre[u] = (cvalid && r0_ + u < rmax) ? (a.tblk ? dK[((col >> 4) * a.N + (rt + r0_ + u)) * 16 + (col & 15)] : dK[(rt + r0_ + u) * a.lddk + col]) : (T)0;
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                const int r = r0_ + u;
                if (r >= rmax) break;
                const int64_t row = rt + r;
                T d[QT], r2 = 0;
                if constexpr (PACKED) {   // float: the q loops two at a time on v_pk_add / v_pk_fma / v_pk_mul_f32
                    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < QT / 2; ++j) {
                        const f32x2 xx = {xs[r * QT + 2 * j], xs[r * QT + 2 * j + 1]};
                        const f32x2 zz = {z[2 * j], z[2 * j + 1]};
                        const f32x2 dd = xx - zz;
                        acc2 = __builtin_elementwise_fma(dd, dd, acc2);
                        d[2 * j] = dd.x; d[2 * j + 1] = dd.y;
                    }
                    r2 = acc2.x + acc2.y;
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) { d[q] = xs[r * QT + q] - z[q]; r2 = fma(d[q], d[q], r2); }
                }
                T k, w;
                cov_and_slope<T, KIND>(r2, k, w);
                T g;
                if (FUSED) {
                    const T t_in = pre[u];
                    T we = 0;
#pragma unroll
                    for (int p = 0; p < PMAX; ++p) if (p < P) we = fma(wsm[r * PMAX + p], e[p], we);
                    qn = fma(k * variance, t_in, qn);
                    g = c1 * ((T)P * t_in + we);
                } else {
                    g = pre[u];
                    if (lower) g = (col < row) ? (T)2 * g : (col == row ? g : (T)0);      // (a select, not a product: the upper triangle may hold NaN bits)
                }
                gvar = fma(g, k, gvar);
                const T W2 = (T)2 * g * w * variance;   // 2 dL/d(r2)
                T* ra = racc + (row - r0) * QA;
                T tq[QT];
                if constexpr (PACKED) {
                    const f32x2 w2 = {W2, W2};
#pragma unroll
                    for (int j = 0; j < QT / 2; ++j) {
                        const f32x2 dd = {d[2 * j], d[2 * j + 1]};
                        const f32x2 t = w2 * dd;
                        f32x2 g2 = {gz[2 * j], gz[2 * j + 1]};
                        f32x2 l2 = {gl[2 * j], gl[2 * j + 1]};
                        g2 = g2 - t;
                        l2 = __builtin_elementwise_fma(-t, dd, l2);
                        gz[2 * j] = g2.x; gz[2 * j + 1] = g2.y;
                        gl[2 * j] = l2.x; gl[2 * j + 1] = l2.y;
                        tq[2 * j] = t.x; tq[2 * j + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        const T t = W2 * d[q];              // dL/d(xs_q) in scaled coordinates
                        gz[q] -= t;
                        gl[q] = fma(-t, d[q], gl[q]);       // dL/dl_q * l_q
                        tq[q] = t;
                    }
                }
                // row side: the QT sums (and, fused, the P sums of R) over this wave's 64 columns: ONE reduce-scatter per 16-lane row
                // (lane l ends with its row's sum of t[l & (QT-1)]), rows folded with the permlane swaps (float), then one LDS atomic
                // instruction from QT (+P) lanes -- the LDS atomic unit, not the VALU, bounds this kernel once the sums are cheap.
                constexpr bool MERGE = FUSED && sizeof(T) == 4 && (QT + PMAX <= 16);
                T rs = 0, rsR = 0;
                if (rowside) rs = row_reduce_scatter<T, QT>(tq, lane);
                if (FUSED && a.R) {
                    const T kv = cvalid ? k * variance : (T)0;
                    T ke[PMAX];
#pragma unroll
                    for (int p = 0; p < PMAX; ++p) ke[p] = kv * e[p];
                    rsR = row_reduce_scatter<T, PMAX>(ke, lane);
                }
                const int l16 = lane & 15;
                if constexpr (MERGE) {
                    const bool isq = l16 < QT;
                    float v = isq ? rs : rsR;
                    v = wave_rows_sum(v);
                    // lane l16 >= QT holds the R sum of index l16 & (PMAX-1) (a permutation of 0..PMAX-1 over lanes QT..QT+PMAX-1)
                    const int rp = l16 & (PMAX - 1);
                    const bool act = isq ? (rowside && l16 < Q) : (a.R != nullptr && l16 < QT + PMAX && rp < P);
                    if (lane < 16 && act) lds_add(ra + (isq ? l16 : QT + rp), v);
                } else if constexpr (sizeof(T) == 4) {
                    if (rowside) { const float v = wave_rows_sum(rs); if (lane < QT && lane < Q) lds_add(ra + lane, v); }
                    if (FUSED && a.R) { const float v = wave_rows_sum(rsR); if (lane < PMAX && lane < P) lds_add(ra + QT + lane, v); }
                } else {
                    if (rowside && l16 < QT && l16 < Q) lds_add(ra + l16, rs);
                    if (FUSED && a.R && l16 < PMAX && l16 < P) lds_add(ra + QT + l16, rsR);
                }
              }
            }
        }
        // column side: stored once when this block owns the whole column; both roles flow into dX in the square case
        T* dXc = a.square ? a.dX : a.dX2;
        const int64_t sXc = a.square ? a.sX : a.sX2;
        if (dXc && cvalid) {
            const bool plain = !a.square && gridDim.y == 1 && (sXc != 0 || gridDim.z == 1);   // this block owns the column
            const T cside = (a.square && a.sym == 1) ? (T)2 : (T)1;
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if (q < Q) {
                    T* p = dXc + (int64_t)s * sXc + col * Q + q;
                    if (plain) *p += gz[q] * il[q]; else atomic_add(p, cside * gz[q] * il[q]);
                }
            }
        }
        if (FUSED) {   // per-sample sums of q_n and |e_n|^2
            const int64_t t1 = (tile0 + 255 < a.N2 - 1) ? tile0 + 255 : a.N2 - 1;
            if (tile0 / a.B == t1 / a.B) {
                const double qs = block_sum<double>((double)qn, redd);
                const double es = (blockIdx.y == 0) ? block_sum<double>(e2, redd) : 0.0;
                if (tid == 0) { atomic_add(a.scal + 2 * (tile0 / a.B), qs); if (blockIdx.y == 0) atomic_add(a.scal + 2 * (tile0 / a.B) + 1, es); }
            } else if (cvalid) {
                atomic_add(a.scal + 2 * (col / a.B), (double)qn);
                if (blockIdx.y == 0) atomic_add(a.scal + 2 * (col / a.B) + 1, e2);
            }
        }
    }
    __syncthreads();
    // row side flush
    if (rowside || (FUSED && a.R)) {
        for (int64_t i = tid; i < (rend - r0) * QA; i += 256) {
            const int64_t r = i / QA;
            const int c = (int)(i % QA);
            if (c < QT) {
                if (rowside && c < Q) atomic_add(a.dX + (int64_t)s * a.sX + (r0 + r) * Q + c, racc[i] / ls[a.ard ? c : 0]);
            } else if (FUSED && a.R && c - QT < P) {
                atomic_add(a.R + (r0 + r) * P + (c - QT), racc[i]);
            }
        }
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

// Generic reverse mode for Q > 16 inputs (the tiled kernel keeps Q values per lane in registers): one thread per (row, column) pair, the
// coordinates re-read from global memory (L2-resident: (N + N2) Q values), row-side sums block-reduced per coordinate.  Correct for any Q,
// far from the tiled kernel's speed -- the counterpart of gram_generic_kernel in gram.hip.  Same accumulate-into semantics.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void gram_bwd_generic_kernel(GramBwdArgs<T> a) {
    __shared__ T red[16];
    const int tid = threadIdx.x;
    const int s = blockIdx.z;
    const int64_t col = (int64_t)blockIdx.x * 256 + tid;
    const bool cvalid = col < a.N2;
    const int Q = a.Q;
    const T* __restrict__ X = a.X + (int64_t)s * a.sX;
    const T* __restrict__ X2 = a.X2 + (int64_t)s * a.sX2;
    const T* __restrict__ ls = a.ls + (int64_t)s * a.sls;
    const T variance = a.var[(int64_t)s * a.svar];
    const T* __restrict__ dK = a.dK + (int64_t)s * a.sdK;
    T* dXc = a.square ? a.dX : a.dX2;
    const int64_t sXc = a.square ? a.sX : a.sX2;
    T gvar = 0, gl0 = 0;
    for (int64_t row = blockIdx.y; row < a.N; row += gridDim.y) {
        T r2 = 0;
        if (cvalid)
            for (int q = 0; q < Q; ++q) { const T d = (X[row * Q + q] - X2[col * Q + q]) / ls[a.ard ? q : 0]; r2 = fma(d, d, r2); }
        T k, w;
        cov_and_slope<T, KIND>(r2, k, w);
        const T g = cvalid ? dK[row * a.lddk + col] : (T)0;
        gvar = fma(g, k, gvar);
        const T W2 = (T)2 * g * w * variance;       // 2 dL/d(r2)
        for (int q = 0; q < Q; ++q) {
            const T il = (T)1 / ls[a.ard ? q : 0];
            const T d = cvalid ? (X[row * Q + q] - X2[col * Q + q]) * il : (T)0;
            const T t = W2 * d;                      // dL/d(scaled x_q)
            if (dXc && cvalid) atomic_add(dXc + (int64_t)s * sXc + col * Q + q, -t * il);
            const T gl = -t * d * il;                // dL/dl_q
            if (a.dX) { const T v = block_sum<T>(t * il, red); if (tid == 0) atomic_add(a.dX + (int64_t)s * a.sX + row * Q + q, v); }
            if (a.dls) {
                if (a.ard) { const T v = block_sum<T>(gl, red); if (tid == 0) atomic_add(a.dls + (int64_t)s * a.sls + q, v); }
                else gl0 += gl;
            }
        }
    }
    if (a.dls && !a.ard) { const T v = block_sum<T>(gl0, red); if (tid == 0) atomic_add(a.dls + (int64_t)s * a.sls, v); }
    if (a.dvar) { const T v = block_sum<T>(gvar, red); if (tid == 0) atomic_add(a.dvar + (int64_t)s * a.svar, v); }
}

// ---- SVGP-fused reverse pass, float32, P = 1, Q <= 8: distances and all cross-lane sums on the matrix pipe --------------------------
// The pass above spends its 93 VALU instructions per pair on the distance (16), the sums over rows (dZ, R: reduce-scatter across the
// wave + an LDS atomic per row) and over columns (dX, dl: 24 multiply-adds).  With W_mn = 2 g w variance (the per-pair weight) and
// scaled coordinates z_m, x_n every one of them is a skinny matrix product:
//   r2_mn = |z_m|^2 + |x_n|^2 - 2 (X Z^T)_nm                                            (stationary.py:98-107, the reference's own form)
//   dZ_mq = (z_mq S_m - B_mq) / l_q,   [B | S] = W   (M x N) . [X | 1] (N x 9)          (S_m = sum_n W_mn)
//   dX_nq = (x_nq C_n - D_nq) / l_q,   [D | C] = W^T (N x M) . [Z | 1] (M x 9)          (C_n = sum_m W_mn)
//   dl_q  = -(sum_m z_mq^2 S_m - 2 sum_m z_mq B_mq + sum_n x_nq^2 C_n) / l_q
// A wave walks 16 (m) x 16 (n) tiles.  Two v_mfma_f32_16x16x4_f32 (true float32) give the tile of dot products in the accumulator layout
// -- lane = (m = l % 16, columns 4 (l / 16) .. + 3) --, which is at once the layout of the T loads (16 bytes per lane) and the A-operand
// layout of the product that contracts over n: [B | S] of the tile's 16 rows accumulates in 4 registers per lane over ALL the columns
// the wave visits.  The tile of W is transposed through 1 KB of LDS (one ds_write_b128 + four ds_read_b32 per lane) and fed to the
// product that contracts over m: [D | C] of the tile's 16 columns, accumulated over the band's rows and flushed per column tile.
// ~15 VALU instructions per pair; ten MFMAs per 256 pairs.
struct BwdMfmaArgs {
    const float* Zs; const float* Xs;    // coordinates / lengthscale, zero-padded to 8 per point (bwd_prescale_kernel)
    const float* Xn;                     // |x_n|^2 of the scaled coordinates
    const float* ls; const float* var; const float* T; const float* U; const float* Y; const float* w;
    const float* noise;
    float* dX; float* dY; double* zacc;  // zacc [M][16]: 0..7 B_mq, 8 S_m, 9 R_m (zeroed by the launcher); float64: ~10^3 workgroups add into it
    double* dls3;                        // [8]: sum_n x_nq^2 C_n
    float* dvar; double* scal;
    int64_t M, SB, B, sY;
    int Q, ard, CT, dY_shared, tblk;     // tblk: T in 16-column blocks, element (m, n) at ((n / 16) * M + m) * 16 + n % 16
    double a1;
    // F16 accumulation (RBF): bit patterns of max |H0| (the T product's A operand), max |w_m|, max |y_n - U_n| -- the bound that scales the weights
    const unsigned* h0max; const unsigned* mx; const unsigned* tmax;     // tmax: max |T| itself when the GEMM reported it (word != 0)
};

#ifndef MXF_MF_MT
#define MXF_MF_MT 8
#endif
constexpr int MF_MT = MXF_MF_MT;             // row tiles of 16 per band: 4 accumulator registers each (16 tiles spill: the allocator chains each
                                     // accumulating MFMA through a second register quad)
constexpr int MF_RB = 16 * MF_MT;    // rows per band
// MFMA chains as ONE inline-asm statement each.  (1) A chain on one accumulator must issue back to back: a single foreign instruction
// between two dependent v_mfma_f32_16x16x4_f32 costs ~43 cycles (MI355X_MICROARCH.md), and the scheduler happily puts v_exp_f32 there.
// (2) In place ("+v"): through the builtin the register allocator chains every accumulation through a second register quad.
// The hazard recogniser does not see these, so every block is hazard-complete by itself: s_nop 4 in front (VALU write -> MFMA read of a
// source register) and s_nop 11 behind (12 wait states >= the 10 an 8-pass MFMA result needs before ANY instruction may touch it -- the
// compiler is free to spill or copy an accumulator right after the block, and did so in the Matern-5/2 instance: wrong dZ until this).
#define MF_DOT2(d, a0, b0, a1, b1)                                                                                      \
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0\n\tv_mfma_f32_16x16x4_f32 %0, %3, %4, %0\n\ts_nop 11"       \
                 : "=&v"(d) : "v"(a0), "v"(b0), "v"(a1), "v"(b1))
// the ten MFMAs of a pipeline stage in ONE block, the three chains (dots of the next tile, row side of this tile, column side of the
// previous tile) interleaved so that no two neighbours share an accumulator: they issue every 32 cycles (a dependent neighbour waits 40)
#define MF_STAGE(d, xa0_, zb0_, xa1_, zb1_, c1, w0, x0, w1, x1, w2, x2, w3, x3, c2, t0, z0, t1, z1, t2, z2, t3, z3)      \
    asm volatile("s_nop 4\n\t"          /* VALU write -> MFMA read of the same VGPR needs wait states the compiler only inserts for builtins */ \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %4, 0\n\t"                                                             \
                 "v_mfma_f32_16x16x4_f32 %1, %7, %8, %1\n\t"                                                            \
                 "v_mfma_f32_16x16x4_f32 %2, %15, %16, %2\n\t"                                                          \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %6, %0\n\t"                                                            \
                 "v_mfma_f32_16x16x4_f32 %1, %9, %10, %1\n\t"                                                           \
                 "v_mfma_f32_16x16x4_f32 %2, %17, %18, %2\n\t"                                                          \
                 "v_mfma_f32_16x16x4_f32 %1, %11, %12, %1\n\t"                                                          \
                 "v_mfma_f32_16x16x4_f32 %2, %19, %20, %2\n\t"                                                          \
                 "v_mfma_f32_16x16x4_f32 %1, %13, %14, %1\n\t"                                                          \
                 "v_mfma_f32_16x16x4_f32 %2, %21, %22, %2\n\ts_nop 11"                                                    \
                 : "=&v"(d), "+v"(c1), "+v"(c2)                                                                          \
                 : "v"(xa0_), "v"(zb0_), "v"(xa1_), "v"(zb1_), "v"(w0), "v"(x0), "v"(w1), "v"(x1), "v"(w2), "v"(x2), "v"(w3), "v"(x3),   \
                   "v"(t0), "v"(z0), "v"(t1), "v"(z1), "v"(t2), "v"(z2), "v"(t3), "v"(z3))
#define MF_ACC4(c, a0, b0, a1, b1, a2, b2, a3, b3)                                                                      \
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x4_f32 %0, %3, %4, %0\n\t"           \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %6, %0\n\tv_mfma_f32_16x16x4_f32 %0, %7, %8, %0\n\ts_nop 11"               \
                 : "+v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3))

// F16 form of a stage (r03): the two ACCUMULATING products contract over the tile's 16 columns / 16 rows, which is exactly the K of one
// v_mfma_f32_16x16x16_f16 -- with the weights and the coordinates split into hi + lo f16 (three products, f32-equivalent as in gemm_split.hip)
// they take 3 + 3 half-length instructions instead of 4 + 4 full-length float32 ones; the dot products stay true float32.
// Chains: d (dots of the next tile), c1 (row side, this tile), c2 (column side, previous tile), interleaved.
#define MF_STAGE16(d, xa0_, zb0_, xa1_, zb1_, c1, wh, wl, xh, xl, c2, th, tl, zh, zl)                                     \
    asm volatile("s_nop 4\n\t"                                                                                            \
                 "v_mfma_f32_16x16x4_f32 %0, %3, %4, 0\n\t"                                                               \
                 "v_mfma_f32_16x16x16_f16 %1, %7, %9, %1\n\t"                                                             \
                 "v_mfma_f32_16x16x16_f16 %2, %11, %13, %2\n\t"                                                           \
                 "v_mfma_f32_16x16x4_f32 %0, %5, %6, %0\n\t"                                                              \
                 "v_mfma_f32_16x16x16_f16 %1, %7, %10, %1\n\t"                                                            \
                 "v_mfma_f32_16x16x16_f16 %2, %11, %14, %2\n\t"                                                           \
                 "v_mfma_f32_16x16x16_f16 %1, %8, %9, %1\n\t"                                                             \
                 "v_mfma_f32_16x16x16_f16 %2, %12, %13, %2\n\ts_nop 11"                                                    \
                 : "=&v"(d), "+v"(c1), "+v"(c2)                                                                            \
                 : "v"(xa0_), "v"(zb0_), "v"(xa1_), "v"(zb1_), "v"(wh), "v"(wl), "v"(xh), "v"(xl), "v"(th), "v"(tl), "v"(zh), "v"(zl))
#define MF_ACC16(c, ah, al, bh, bl)                                                                                       \
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x16_f16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x16_f16 %0, %1, %4, %0\n\t"        \
                 "v_mfma_f32_16x16x16_f16 %0, %2, %3, %0\n\ts_nop 11"                                                      \
                 : "+v"(c) : "v"(ah), "v"(al), "v"(bh), "v"(bl))

#ifdef MXF_BWD_TRACE
// probe build only (tests/probes/bwd_trace.py): shader-clock stamps of the stages of one wave's row tiles, [column tile it < 8][row tile][stage]
__device__ unsigned bwd_trace_buf[8 * 8 * 8];
extern "C" int mxf_debug_bwd_trace(unsigned* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bwd_trace_buf), sizeof(bwd_trace_buf)); }
#define BT_STAMP2(mtv, k, dep)                                                                                            \
    do {                                                                                                                  \
        unsigned long long t_;                                                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                            \
        if (traced && it < 8 && lane == 0) bwd_trace_buf[(it * 8 + (mtv)) * 8 + (k)] = (unsigned)t_;                      \
    } while (0)
#define BT_STAMP(k, dep)                                                                                                  \
    do {                                                                                                                  \
        unsigned long long t_;                                                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                            \
        if (traced && it < 8 && lane == 0) bwd_trace_buf[(it * 8 + mt) * 8 + (k)] = (unsigned)t_;                         \
    } while (0)
#else
#define BT_STAMP(k, dep) do { } while (0)
#define BT_STAMP2(mtv, k, dep) do { } while (0)
#endif
typedef _Float16 bw_f16x4 __attribute__((ext_vector_type(4)));
// x (4 floats) = hi + lo, f16 each (hi = round(x), lo = round(x - hi)): the A / B operand of v_mfma_f32_16x16x16_f16 (k = 4 (lane / 16) + i)
__device__ __forceinline__ void split4(const float (&x)[4], bw_f16x4& hi, bw_f16x4& lo) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = {x[0], x[1], x[2], x[3]};
    hi = __builtin_convertvector(v, bw_f16x4);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bw_f16x4);
}

// LDS-DMA requests of the DMAT form below (inline asm: the compiler neither tracks nor waits for them -- the waits are counted by hand)
//   bw_dma16: 64 lanes x 16 bytes from sbase + voff (wave-uniform base in SGPRs, per-lane byte offset) to the KB at LDS byte address lds
//   bw_dma4 / bw_dma4v: 64 lanes x 4 bytes to the 256 bytes at lds; the v form takes a full per-lane address
__device__ __forceinline__ void bw_dma16(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void bw_dma4(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void bw_dma4v(const float* p, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p), "s"(lds) : "memory", "m0");
}

// DMAT (r06; RBF, FULL, F16, T in 16-column blocks): the T tiles and the column-side values reach the wave through LDS-DMA instead of
// registers.  The register form keeps PD = 4 tiles (64 bytes) per lane in flight -- at two waves per SIMD 32 KB per CU, about half of what
// 8 TB/s x the loaded memory latency asks for -- and has no registers for more (256 VGPRs).  Here every wave owns a ring of eight 1 KB
// slots, slot = row tile: at row tile mt the request of the tile that will next use the slot just consumed is issued (tile 7 of this
// column tile at mt = 0, tile mt - 1 of the NEXT column tile otherwise), i.e. seven tiles = 7 KB per wave are in flight, 56 KB per CU.
// A tile is one contiguous KB of the blocked T, so its LDS image is the memory image and lane (li, lq) reads its four values at
// 64 li + 16 lq.  The sixteen columns' x, |x|^2, U and y of the next column tile come the same way (three requests, double-buffered).
// Waits are counted: requests retire in order, so "all but the youngest N" is exact -- N = the requests issued behind the one needed
// (the compiler's own memory operations -- the dX atomics of the previous column tile -- are counted when they are known to be there,
// anything uncounted only makes a wait stricter).
template <int KIND, bool FULL, bool F16, bool DMAT = false>       // FULL: M % MF_RB == 0 and SB % 64 == 0 (no ragged tiles: no masks); F16: RBF only
__global__ __launch_bounds__(256, MXF_MF_MT <= 4 ? 3 : 2) void svgp_bwd_mfma_kernel(BwdMfmaArgs a) {
    static_assert(!F16 || KIND == MXF_K_RBF, "the f16 accumulation is scaled for the RBF weights");
    static_assert(!DMAT || (FULL && F16 && MF_MT == 8), "the LDS-DMA form: full tiles, f16 accumulation, eight row tiles (slot = row tile)");
    __shared__ __attribute__((aligned(16))) float tring[DMAT ? 4 : 1][DMAT ? 8 * 256 : 4];       // per wave: eight T tiles
    __shared__ __attribute__((aligned(16))) float cbuf[DMAT ? 4 : 1][2][DMAT ? 192 : 4];        // per wave, two buffers: [x 16 x 8 | |x|^2 16 | U 16 | y 16 | -]
    constexpr int QT = 8;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // LDS tables of the band (bank-conflict free for the access patterns below: PMC showed half of the LDS cycles in conflicts before)
    __shared__ __attribute__((aligned(16))) float za[F16 ? 1 : MF_RB][16];     // [z (scaled, 8) | 1 | 0 ...]: B operand of the column-side product (row-contiguous reads)
    // F16: the same table as hi / lo f16, four consecutive rows of one entry j in 8 bytes: [row / 4][j][row % 4] -- the B operand of the 16 x 16 x 16 product
    __shared__ __attribute__((aligned(16))) bw_f16x4 zah[F16 ? MF_RB / 4 : 1][16], zal[F16 ? MF_RB / 4 : 1][16];
    __shared__ __attribute__((aligned(16))) float zd[MF_RB][12];     // [z (8) | |z|^2 | w | - | -]: lanes read (row li, word lq): 12-word rows keep 16 rows x 4 words apart
    __shared__ float rowacc[MF_RB][10];
    __shared__ __attribute__((aligned(16))) float wt[4][2][16][20];  // per wave, double-buffered: the W tile, transposed on the way through (20-word rows)
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = DMAT ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6, li = lane & 15, lq = lane >> 4;
    const int64_t band0 = (int64_t)blockIdx.y * MF_RB;
    const int Q = a.Q;
    const float* __restrict__ Xs = a.Xs;
    const float* __restrict__ Tm = a.T;
    const float ilj = (li < Q) ? 1.f / a.ls[a.ard ? li : 0] : 0.f;                 // 1 / l of coordinate j = li (column flush)
    constexpr bool RX = KIND == MXF_K_RBF;
    constexpr float CS = RX ? 0.84932180028801904272f : 1.f;        // the coordinates' extra scale (bwd_prescale_kernel): sqrt(log2(e) / 2) for RBF
    const float variance = a.var[0];
    const float c1 = (float)a.a1 / a.noise[0];
    const float kc = -c1 * variance;
    if constexpr (F16) {
        for (int i = tid; i < (MF_RB / 4) * 16; i += 256) {
            const int g4 = i / 16, j = i % 16;
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t r = band0 + 4 * g4 + t;
                v[t] = (r < a.M) ? ((j < QT) ? a.Zs[r * QT + j] : (j == 8 ? 1.f : 0.f)) : 0.f;
            }
            split4(v, zah[g4][j], zal[g4][j]);
        }
    } else {
        for (int i = tid; i < MF_RB * 16; i += 256) {
            const int r = i / 16, j = i % 16;
            za[r][j] = (band0 + r < a.M) ? ((j < QT) ? a.Zs[(band0 + r) * QT + j] : (j == 8 ? 1.f : 0.f)) : 0.f;
        }
    }
    // F16: weights are accumulated as (u k) 2^esc, |u k| 2^esc <= 2^14 from max |T| (reported by the split GEMM; else the bound
    // variance M max|H0|) and |w e| <= max|w| max|e|
    // (k <= 1 up to rounding); -(c1 variance) 2^-esc is applied when the sums are flushed.  f16 subnormals are kept by the matrix pipe
    // (tests/probes/probe_mfma_overlap.hip), so a loose bound costs absolute, not relative, precision: 2^-25 / 2^14 of the bound.
    float escf = 0.f, unsc = 1.f, fl = 1.f;     // fl: what the f16 sums still lack, -(c1 variance) 2^-esc
    if constexpr (F16) {
        const unsigned tb = a.tmax ? a.tmax[0] : 0u;
        const float bnd = (tb ? __builtin_bit_cast(float, tb) : variance * (float)a.M * __builtin_bit_cast(float, a.h0max[0])) +
                          __builtin_bit_cast(float, a.mx[0]) * __builtin_bit_cast(float, a.mx[1]);
        const unsigned bb = __builtin_bit_cast(unsigned, bnd);
        const int ex = (int)((bb >> 23) & 0xff);
        int esc = (ex == 0 || ex == 0xff) ? 0 : 13 - (ex - 127);          // bnd < 2^(ex - 126): bnd 2^esc < 2^14
        esc = esc > 60 ? 60 : (esc < -60 ? -60 : esc);
        escf = (float)esc;
        unsc = __builtin_bit_cast(float, (unsigned)(127 - esc) << 23);    // 2^-esc
        fl = kc * unsc;
    }
    for (int r = tid; r < MF_RB; r += 256) {
        float n2 = 0.f;
        const bool ok = band0 + r < a.M;
#pragma unroll
        for (int q = 0; q < QT; ++q) { const float v = ok ? a.Zs[(band0 + r) * QT + q] : 0.f; zd[r][q] = v; n2 = fmaf(v, v, n2); }
        zd[r][8] = n2;
        zd[r][9] = ok ? a.w[band0 + r] : 0.f;
        zd[r][10] = 0.f; zd[r][11] = 0.f;
    }
    for (int i = tid; i < MF_RB * 10; i += 256) (&rowacc[0][0])[i] = 0.f;
    __syncthreads();

    f32x4 C1[MF_MT];
    float racc[MF_MT];
#pragma unroll
    for (int mt = 0; mt < MF_MT; ++mt) { C1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; racc[mt] = 0.f; }
    float gvar = 0.f, dl3 = 0.f;
    double qsum = 0.0, esum = 0.0;
    int cur_s = -1;
    auto flush_scal = [&]() {       // wave-uniform call: per-sample sums of q_n (and |e_n|^2 from the first band)
        const double qs = wave_sum(qsum), es = wave_sum(esum);
        if (lane == 0 && cur_s >= 0) { atomic_add(a.scal + 2 * cur_s, qs); if (blockIdx.y == 0) atomic_add(a.scal + 2 * cur_s + 1, es); }
        qsum = 0.0; esum = 0.0;
    };
    float* wtw = &wt[wave][0][0][0];
    constexpr int WTB = 16 * 20;      // words per transpose buffer
    _Float16* const wth = reinterpret_cast<_Float16*>(wtw);      // F16: the same space as two planes of 16 x 20 halves per buffer
    constexpr int WTB16 = 2 * WTB;    // halves per transpose buffer
    // row of T this lane reads in row tile mt: band0 + 16 mt + li (clamped: ragged rows are masked, not skipped -- no branches around loads)
    const int64_t rowl = band0 + li;

    // Loads run AHEAD of their use across the column tiles (r03): the raw column-side values of tile it + 1 are requested at the top of
    // tile it, and the T tiles PD row tiles ahead -- from tile it + 1's first rows while tile it works on its last ones.  (Before, every
    // column tile began with a round of loads that were used at once and with T only two row tiles ahead: PMC had the waves waiting on
    // memory for 42 % of their cycles.)
    // (the Matern instances have no registers to spare -- they spill with the deeper pipeline: T two tiles ahead, the next column tile's
    //  values requested behind the row tiles instead of in front of them)
    constexpr bool AHEAD = RX;
    constexpr int PD = AHEAD ? 4 : 2;                      // T tiles in flight per lane (MF_MT % PD == 0: a tile's slot is its index mod PD)
    static_assert(MF_MT % PD == 0, "slot = row tile % PD needs MF_MT % PD == 0");
    const int64_t tstep = a.tblk ? 256 : 16 * a.SB;       // blocked: a wave's 16 x 16 tile is ONE contiguous KB, the next row tile the next KB
    const int sb = (int)a.SB, bsz = (int)a.B;
    struct Cols { float xa0, xa1; f32x4 xx, uu, yy; float xv[4]; int smp; };
    auto col_nt0 = [&](int it_) -> int { return ((int)blockIdx.x * a.CT + it_) * 64 + wave * 16; };   // (SB < 2^31: the launcher checks)   // the wave's 16 columns (the block's four waves side by side)
    // (the sample of a column tile, nt0 / B, is walked along: a wave's tiles are 64 columns apart, and a 64-bit division per tile is ~100 instructions)
    auto load_cols = [&](int nt0_, int smp_) -> Cols {
        Cols c;
        c.smp = smp_;                                                                // B % 16 == 0: a tile lies inside one sample
        const int n0_ = nt0_ + 4 * lq;
        const int64_t n0c_ = (FULL || n0_ < sb) ? n0_ : sb - 4;
        const int64_t nac_ = (FULL || nt0_ + li < sb) ? nt0_ + li : sb - 1;     // column of the dot product's A operand
        c.xa0 = Xs[nac_ * QT + lq]; c.xa1 = Xs[nac_ * QT + 4 + lq];
        c.xx = *reinterpret_cast<const f32x4*>(a.Xn + n0c_);
        c.uu = *reinterpret_cast<const f32x4*>(a.U + n0c_);
        c.yy = *reinterpret_cast<const f32x4*>(a.Y + (int64_t)c.smp * a.sY + (n0c_ - (int64_t)c.smp * a.B));     // (16-byte aligned: B % 16 == 0, sY = 0 or B)
#pragma unroll
        for (int t = 0; t < 4; ++t) c.xv[t] = Xs[(n0c_ + t) * QT + (li & 7)];
        return c;
    };
    // T rows of this lane: band0 + li + 16 mt; ragged bands clamp to the last row and mask the value instead of branching
    auto tbase = [&](int nt0_, const float*& tl_) -> const float* {
        const int n0_ = nt0_ + 4 * lq;
        const int64_t n0c_ = (FULL || n0_ < sb) ? n0_ : sb - 4;
        tl_ = a.tblk ? Tm + ((int64_t)(nt0_ >> 4) * a.M + a.M - 1) * 16 + 4 * lq : Tm + n0c_ + (a.M - 1) * a.SB;
        return a.tblk ? Tm + ((int64_t)(nt0_ >> 4) * a.M + rowl) * 16 + 4 * lq : Tm + n0c_ + rowl * a.SB;
    };
    auto tget = [&](const float* base_, const float* tl_, int j) -> f32x4 {
        const float* q = base_ + (int64_t)j * tstep;
        if (!FULL) q = q <= tl_ ? q : tl_;
        return *reinterpret_cast<const f32x4*>(q);
    };
    int nt0 = col_nt0(0);
#ifdef MXF_BWD_TRACE
    const bool traced = blockIdx.x == gridDim.x / 2 && blockIdx.y == 3 && wave == 1;
#endif
    if (nt0 < sb) {
    Cols cur;
    const float* tl_c = nullptr;
    const float* tb_c = nullptr;
    f32x4 tq[PD];
    int smp_w = nt0 / bsz, smp_end = (smp_w + 1) * bsz;       // sample of the tile at nt0 and the first column behind it (< 2^31 + B: unsigned compare)
    auto advance_smp = [&](int nt0_) { while ((unsigned)nt0_ >= (unsigned)smp_end) { ++smp_w; smp_end += bsz; } return smp_w; };
    // DMAT: request side
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const unsigned lds_t = DMAT ? __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)&tring[DMAT ? wave : 0][0]) : 0u;
    const unsigned lds_c = DMAT ? __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)&cbuf[DMAT ? wave : 0][0][0]) : 0u;
    const unsigned voff16 = (unsigned)lane * 16u, voff4 = (unsigned)lane * 4u;
    const bool hasdx = a.dX != nullptr;
    auto tile_base = [&](int nt0_) -> const float* {              // the wave's column block of T at the band's first row (wave-uniform: SGPRs)
        const int64_t off = ((int64_t)(nt0_ >> 4) * a.M + band0) * 16;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)off), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)off >> 32));
        return Tm + (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto issue_cols = [&](int nt0_, int smp_, int buf) {
        const unsigned nts = __builtin_amdgcn_readfirstlane((unsigned)nt0_);
        const float* xb = Xs + (int64_t)nts * QT;
        bw_dma4(xb, voff4, lds_c + buf * 768);
        bw_dma4(xb, voff4 + 256u, lds_c + buf * 768 + 256);
        // lanes 0..15 |x|^2, 16..31 U, 32..47 y, 48..63 |x|^2 again (padding)
        const float* p = (lq == 1 ? a.U + nt0_ : lq == 2 ? a.Y + (int64_t)smp_ * a.sY + (nt0_ - (int64_t)smp_ * a.B) : a.Xn + nt0_) + li;
        bw_dma4v(p, lds_c + buf * 768 + 512);
    };
    auto read_cols = [&](int buf, int smp_) -> Cols {
        Cols c;
        const float* cb = &cbuf[DMAT ? wave : 0][DMAT ? buf : 0][0];
        c.smp = smp_;
        c.xa0 = cb[li * 8 + lq]; c.xa1 = cb[li * 8 + 4 + lq];
        c.xx = *reinterpret_cast<const f32x4*>(cb + 128 + 4 * lq);
        c.uu = *reinterpret_cast<const f32x4*>(cb + 144 + 4 * lq);
        c.yy = *reinterpret_cast<const f32x4*>(cb + 160 + 4 * lq);
#pragma unroll
        for (int t = 0; t < 4; ++t) c.xv[t] = cb[(4 * lq + t) * 8 + (li & 7)];
        return c;
    };
    if constexpr (DMAT) {
        issue_cols(nt0, smp_w, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur = read_cols(0, smp_w);
        tb_c = tile_base(nt0);
#pragma unroll
        for (int j = 0; j < 7; ++j) bw_dma16(tb_c + j * 256, voff16, lds_t + j * 1024);
    } else if constexpr (AHEAD) {
        cur = load_cols(nt0, smp_w);
        tb_c = tbase(nt0, tl_c);
#pragma unroll
        for (int j = 0; j < PD; ++j) tq[j] = tget(tb_c, tl_c, j);
    }
    for (int it = 0; it < a.CT; ++it) {
        int nt0n = nt0;
        bool has_next = false;
        Cols nxt;
        const float* tl_n = tl_c;
        const float* tb_n = tb_c;
        int smp_n = cur.smp;
        if constexpr (DMAT) {
            nt0n = col_nt0(it + 1);
            has_next = it + 1 < a.CT && nt0n < sb;
            if (!has_next) nt0n = nt0;                        // (no next tile: harmless requests of this one again -- the counts stay the same)
            smp_n = has_next ? advance_smp(nt0n) : cur.smp;
            issue_cols(nt0n, smp_n, (it + 1) & 1);
            tb_n = tile_base(nt0n);
        } else if constexpr (!AHEAD) {         // everything this column tile needs is requested here, at its top
            cur = load_cols(nt0, advance_smp(nt0));
            tb_c = tbase(nt0, tl_c);
#pragma unroll
            for (int j = 0; j < PD; ++j) tq[j] = tget(tb_c, tl_c, j);
        } else {
            nt0n = col_nt0(it + 1);
            has_next = it + 1 < a.CT && nt0n < sb;
            if (!has_next) nt0n = nt0;                        // (no next tile: harmless re-loads of this one)
            nxt = load_cols(nt0n, has_next ? advance_smp(nt0n) : cur.smp);
            tb_n = tbase(nt0n, tl_n);
        }
        BT_STAMP2(0, 6, tb_n);
        const int smp = cur.smp;
        if (smp != cur_s) { flush_scal(); cur_s = smp; }
        const int n0 = nt0 + 4 * lq;                                                 // this lane's 4 consecutive columns
        const bool cval = FULL || n0 < sb;                                         // SB % 4 == 0: all four or none
        const float xa0 = cur.xa0, xa1 = cur.xa1;
        const f32x4 xx = cur.xx;
        float e[4], bx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bx[t] = cval ? ((li < QT) ? cur.xv[t] : (li == 8 ? 1.f : 0.f)) : 0.f;
            e[t] = cval ? cur.yy[t] - cur.uu[t] : 0.f;
        }
        if (blockIdx.y == 0 && li == 0 && cval) {          // one lane per column, first band only: dY and |e|^2
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                esum += (double)e[t] * (double)e[t];
                if (a.dY) {
                    const float g = -c1 * e[t];
                    const int64_t n = n0 + t;
                    if (a.dY_shared) atomic_add(a.dY + (n - (int64_t)smp * a.B), g); else a.dY[n] = g;
                }
            }
        }
        bw_f16x4 bxh, bxl;
        f32x4 xxs = xx;
        if constexpr (F16) { split4(bx, bxh, bxl); xxs = xx - escf; }       // r2 - esc: k comes out as k 2^esc
        BT_STAMP2(0, 7, xxs[0]);
        float qn = 0.f;
        f32x4 C2 = f32x4{0.f, 0.f, 0.f, 0.f};
        // software pipeline over the row tiles: the dot products of tile mt + 1 are issued BEFORE the arithmetic of tile mt (whose dots were
        // issued one iteration earlier), and the accumulating products of tile mt / the transposed product of tile mt - 1 AFTER it -- so the
        // matrix pipe works on ten MFMAs while the VALU does the next tile, instead of the two taking turns
        // (tile 0's dots through the builtin: their first use follows at once, and only the builtin tells the hazard recogniser)
        f32x4 dotc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0, zd[li][lq], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dotc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1, zd[li][4 + lq], dotc, 0, 0, 0);
        f32x2 zwc = *reinterpret_cast<const f32x2*>(&zd[li][8]);
        float zbn0 = zd[(1 < MF_MT ? 16 : 0) + li][lq], zbn1 = zd[(1 < MF_MT ? 16 : 0) + li][4 + lq];
        f32x2 zwn = *reinterpret_cast<const f32x2*>(&zd[(1 < MF_MT ? 16 : 0) + li][8]);
#pragma unroll
        for (int mt = 0; mt < MF_MT; ++mt) {
            const int rl = mt * 16 + li;
            asm volatile("" ::: "memory");
            BT_STAMP(0, rl);
            f32x4 tv;
            if constexpr (DMAT) {
                // request into the slot consumed one row tile ago, then wait for this row tile's slot: requests behind it = the six others
                // of the ring + this one + the three column requests issued at the top (mt < 7; at mt = 7 the slot was requested at mt = 0,
                // seven tile requests ago) + the four dX atomics of the previous column tile when the caller wants dX
                if (mt == 0) bw_dma16(tb_c + 7 * 256, voff16, lds_t + 7 * 1024);
                else bw_dma16(tb_n + (mt - 1) * 256, voff16, lds_t + (mt - 1) * 1024);
                if (mt == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else if (hasdx && it > 0) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                tv = *reinterpret_cast<const f32x4*>(&tring[DMAT ? wave : 0][DMAT ? mt * 256 + li * 16 + 4 * lq : 0]);
            } else {
            tv = tq[mt % PD];
            BT_STAMP(1, tv[0]);
            if (mt + PD < MF_MT) tq[mt % PD] = tget(tb_c, tl_c, mt + PD);                               // PD row tiles ahead
            else if constexpr (AHEAD) tq[mt % PD] = tget(tb_n, tl_n, mt + PD - MF_MT);                  // ... into the next column tile
            }
            if (!FULL) { const bool ok = cval && rowl + 16 * mt < a.M; tv = ok ? tv : f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 dotn = dotc;
            const f32x2 zwv = zwc;
            zwc = zwn;
            const float zbc0 = zbn0, zbc1 = zbn1;                   // operands of the NEXT tile's dot products (this stage's MFMA block)
            if (mt + 2 < MF_MT) {                                   // ... and those two tiles ahead, |z|^2 and w
                zbn0 = zd[rl + 32][lq]; zbn1 = zd[rl + 32][4 + lq];
                zwn = *reinterpret_cast<const f32x2*>(&zd[rl + 32][8]);
            }
            // the transposed copy of the PREVIOUS row tile (written one iteration ago)
            float wtr[4], zb2[4];
            bw_f16x4 th, tl, zh, zl;
            if (mt > 0) {
                if constexpr (F16) {        // the product's k index is the row 4 lq + t: four consecutive rows of column li, 8 bytes per plane
                    th = *reinterpret_cast<const bw_f16x4*>(wth + ((mt - 1) & 1) * WTB16 + li * 20 + 4 * lq);
                    tl = *reinterpret_cast<const bw_f16x4*>(wth + ((mt - 1) & 1) * WTB16 + 320 + li * 20 + 4 * lq);
                    zh = zah[(mt - 1) * 4 + lq][li]; zl = zal[(mt - 1) * 4 + lq][li];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        wtr[t] = wtw[((mt - 1) & 1) * WTB + (lq + 4 * t) * 20 + li];
                        zb2[t] = za[(mt - 1) * 16 + lq + 4 * t][li];
                    }
                }
            }
            const float wm = zwv[1];
            f32x4 W;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float r2 = fmaf(-2.f, dotc[t], zwv[0] + xxs[t]);
                if constexpr (F16) {
                    const float k = __builtin_amdgcn_exp2f(-r2);          // k 2^esc
                    const float u = fmaf(wm, e[t], tv[t]);
                    W[t] = u * k;
                    qn = fmaf(k, tv[t], qn);
                    racc[mt] = fmaf(k, e[t], racc[mt]);
                } else if constexpr (RX) {
                    // RBF: coordinates carry sqrt(log2(e) / 2) (as the forward Gram kernels'), so k = 2^-r2 is the bare v_exp_f32; the weight is
                    // W = 2 g w variance = -(c1 variance) (T + w e) k; variance and the sum over pairs of g k are applied once, at the flush
                    const float k = __builtin_amdgcn_exp2f(-r2);          // (r2 may round a few ulps below 0 for coincident points: k = 1 + O(1e-6))
                    const float u = fmaf(wm, e[t], tv[t]);
                    W[t] = kc * (u * k);
                    qn = fmaf(k, tv[t], qn);
                    racc[mt] = fmaf(k, e[t], racc[mt]);
                } else {
                    r2 = r2 > 0.f ? r2 : 0.f;
                    float k, w;
                    cov_and_slope<float, KIND>(r2, k, w);
                    const float kv = k * variance;
                    const float g = c1 * fmaf(wm, e[t], tv[t]);
                    W[t] = 2.f * g * w * variance;
                    gvar = fmaf(g, k, gvar);
                    qn = fmaf(kv, tv[t], qn);
                    racc[mt] = fmaf(kv, e[t], racc[mt]);
                }
            }
            // this stage's MFMAs: dots of tile mt + 1, [B | S] += W . [X | 1] of tile mt, [D | C] += W^T . [Z | 1] of tile mt - 1
            bw_f16x4 wh16, wl16;
            BT_STAMP(2, W[3]);
            if constexpr (F16) {
                const float wf[4] = {W[0], W[1], W[2], W[3]};
                split4(wf, wh16, wl16);
                BT_STAMP(3, wl16);
                const bw_f16x4 wh = wh16, wl = wl16;
                if (mt > 0 && mt + 1 < MF_MT) {
                    MF_STAGE16(dotn, xa0, zbc0, xa1, zbc1, C1[mt], wh, wl, bxh, bxl, C2, th, tl, zh, zl);
                } else {
                    if (mt + 1 < MF_MT) MF_DOT2(dotn, xa0, zbc0, xa1, zbc1);
                    MF_ACC16(C1[mt], wh, wl, bxh, bxl);
                    if (mt > 0) MF_ACC16(C2, th, tl, zh, zl);
                }
            } else if (mt > 0 && mt + 1 < MF_MT) {
                MF_STAGE(dotn, xa0, zbc0, xa1, zbc1, C1[mt], W[0], bx[0], W[1], bx[1], W[2], bx[2], W[3], bx[3],
                         C2, wtr[0], zb2[0], wtr[1], zb2[1], wtr[2], zb2[2], wtr[3], zb2[3]);
            } else {
                if (mt + 1 < MF_MT) MF_DOT2(dotn, xa0, zbc0, xa1, zbc1);
                MF_ACC4(C1[mt], W[0], bx[0], W[1], bx[1], W[2], bx[2], W[3], bx[3]);
                if (mt > 0) MF_ACC4(C2, wtr[0], zb2[0], wtr[1], zb2[1], wtr[2], zb2[2], wtr[3], zb2[3]);
            }
            BT_STAMP(4, C1[mt][0]);
            // transpose the tile through LDS: written as (m = li, n = 4 lq .. + 3), read (next iteration) as (n = li, m = lq + 4 t)
            __builtin_amdgcn_wave_barrier();
            if constexpr (F16) {
                // the hi / lo planes go through LDS already split, element (m = li, n = 4 lq + t) to [n][m]: eight 2-byte stores (20-half rows:
                // the four column groups of one store land 32 bytes apart, conflict-free), read back as 8 bytes per plane -- the weights are
                // converted once for both products
                _Float16* const wb = wth + (mt & 1) * WTB16;
#pragma unroll
                for (int t = 0; t < 4; ++t) { wb[(4 * lq + t) * 20 + li] = wh16[t]; wb[320 + (4 * lq + t) * 20 + li] = wl16[t]; }
            } else {
                *reinterpret_cast<f32x4*>(wtw + (mt & 1) * WTB + li * 20 + 4 * lq) = W;
            }
            __builtin_amdgcn_wave_barrier();
            dotc = dotn;
            BT_STAMP(5, dotc[0]);
            __builtin_amdgcn_sched_barrier(0);      // keep the unrolled row tiles apart
        }
        {   // the last row tile's transposed product
            float wtr[4], zb2[4];
            if constexpr (F16) {
                const bw_f16x4 th = *reinterpret_cast<const bw_f16x4*>(wth + ((MF_MT - 1) & 1) * WTB16 + li * 20 + 4 * lq);
                const bw_f16x4 tl = *reinterpret_cast<const bw_f16x4*>(wth + ((MF_MT - 1) & 1) * WTB16 + 320 + li * 20 + 4 * lq);
                const bw_f16x4 zh = zah[(MF_MT - 1) * 4 + lq][li], zl = zal[(MF_MT - 1) * 4 + lq][li];
                MF_ACC16(C2, th, tl, zh, zl);
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                wtr[t] = wtw[((MF_MT - 1) & 1) * WTB + (lq + 4 * t) * 20 + li];
                zb2[t] = za[(MF_MT - 1) * 16 + lq + 4 * t][li];
            }
            MF_ACC4(C2, wtr[0], zb2[0], wtr[1], zb2[1], wtr[2], zb2[2], wtr[3], zb2[3]);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // inline-asm MFMA result -> VALU read: the hazard recogniser does not see it
        }
        BT_STAMP2(7, 6, C2[0]);
        qsum += (double)(RX ? qn * (variance * unsc) : qn);
        // column side: C2[r] = [D | C] of column nt0 + 4 lq + r (= this lane's column n0 + r), entry j = li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float Cn = __shfl(C2[r], (lane & 48) | 8, 64);
            const float pr = bx[r] * Cn;                       // x_nq C_n (q = li; bx is x of column n0 + r at coordinate li)
            if (li < Q) {
                dl3 = fmaf(bx[r], pr, dl3);
                if (a.dX && cval) atomic_add(a.dX + (int64_t)(n0 + r) * Q + li, (pr - C2[r]) * (ilj * (1.f / CS) * fl));
            }
        }
        BT_STAMP2(7, 7, dl3);
        if constexpr (DMAT) {
            if (!has_next) break;
            cur = read_cols((it + 1) & 1, smp_n);     // (landed: the wait of row tile 7 covers everything but the seven youngest requests)
            nt0 = nt0n; tb_c = tb_n;
        } else if constexpr (AHEAD) {
            if (!has_next) break;
            cur = nxt; nt0 = nt0n; tb_c = tb_n; tl_c = tl_n;
        } else {
            nt0 = col_nt0(it + 1);
            if (it + 1 >= a.CT || nt0 >= sb) break;
        }
    }
    if constexpr (DMAT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no request may land in LDS after the workgroup has gone
    }
    flush_scal();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // row side: C1[mt][r] = [B | S] of row band0 + 16 mt + 4 lq + r, column li; the block's four waves are combined in LDS first
#pragma unroll
    for (int mt = 0; mt < MF_MT; ++mt) {
        if (li < 9) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lds_add(&rowacc[mt * 16 + 4 * lq + r][li], C1[mt][r] * fl);
        }
        float rr = racc[mt];                                   // R partials of row 16 mt + li: fold the four column groups
        rr += __shfl_xor(rr, 16, 64);
        rr += __shfl_xor(rr, 32, 64);
        if (lane < 16) lds_add(&rowacc[mt * 16 + li][9], RX ? rr * (variance * unsc) : rr);
    }
    __syncthreads();
    for (int i = tid; i < MF_RB * 10; i += 256) {
        const int r = i / 10, c = i % 10;
        if (band0 + r < a.M) atomic_add(a.zacc + (band0 + r) * 16 + c, (double)rowacc[r][c]);
    }
    if (a.dvar && !RX) { const float v = block_sum<float>(gvar, red); if (tid == 0) atomic_add(a.dvar, v); }     // RBF: -(sum of S) / variance, by the finishing kernel
    {   // sum_n x_nq^2 C_n: lane (q = li) holds its share
        float v = (li < Q) ? dl3 * fl : 0.f;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16 && li < Q) atomic_add(a.dls3 + li, (double)v);       // in the kernel's coordinates: the finishing kernel divides by CS^2
    }
}

// dZ, dls, R from the row-side sums of svgp_bwd_mfma_kernel (one row per thread; float64: z^2 S - 2 z B + x^2 C cancels a digit or two)
// cs: the extra scale of the pass's coordinates (sqrt(log2(e) / 2) for RBF, whose pass also leaves dvar = -(sum_m S_m) / variance to this kernel)
__global__ __launch_bounds__(256) void svgp_bwd_finish_kernel(int64_t M, int Q, int ard, const float* __restrict__ Z /* prescaled: Zs */, const float* __restrict__ ls,
                                                              const double* __restrict__ zacc, const double* __restrict__ dls3,
                                                              float* __restrict__ dZ, float* __restrict__ dls, float* __restrict__ R, double cs,
                                                              const float* __restrict__ var, float* __restrict__ dvar_from_S) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * 256 + tid;
    double g12[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g12[q] = 0.0;
    double Srow = 0.0;
    if (m < M) {
        const double S = zacc[m * 16 + 8];
        Srow = S;
        if (R) R[m] += (float)zacc[m * 16 + 9];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < Q) {
                const double ilq = 1.0 / (double)ls[ard ? q : 0];
                const double z = (double)Z[m * 8 + q], Bq = zacc[m * 16 + q];      // the pass's own scaled, centred coordinate (bwd_prescale_kernel's output: 8 per row)
                if (dZ) dZ[m * Q + q] += (float)((z * S - Bq) * ilq / cs);
                g12[q] = z * (z * S - 2.0 * Bq);
            }
        }
    }
    if (dvar_from_S) { const double v = block_sum<double>(Srow, red); if (tid == 0) atomic_add(dvar_from_S, (float)(-v / (double)var[0])); }
    if (!dls) return;
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q >= Q) break;
        const double v = block_sum<double>(g12[q], red);
        if (tid == 0) {
            const double glq = -(v + (blockIdx.x == 0 ? dls3[q] : 0.0)) / (cs * cs);   // sum over pairs of -W d_q^2 (this block's rows; block 0 adds the column term)
            if (ard) atomic_add(dls + q, (float)(glq / (double)ls[q])); else tot += glq;
        }
    }
    if (tid == 0 && !ard) atomic_add(dls, (float)(tot / (double)ls[0]));
}

template <typename T, int QT, int KIND, int PT>
int launch_q(mxf_ctx* h, GramBwdArgs<T> a, int S, hipStream_t st) {
    constexpr bool FUSED = PT > 0;
    constexpr int QA = QT + PT;
    const size_t fixed = (size_t)(TRB * QT + TRB * PT + 16) * sizeof(T) + 16 * sizeof(double) + 64;
    // LDS per block (the row accumulators): 30 KB = five blocks per CU; 80 KB (two blocks, fewer row flushes) measured 1 % slower per step
    static const int bud_env = MXF_KNOB("MXF_BWD_LDS_KB", 30);
    const size_t budget = (size_t)bud_env * 1024;
    int64_t rb = (int64_t)((budget - fixed) / (QA * sizeof(T)));
    rb = rb / TRB * TRB;
    if (rb < TRB) rb = TRB;
    const int64_t npad = (a.N + TRB - 1) / TRB * TRB;
    if (rb > npad) rb = npad;
    {   // small problems (the M x M core Gram): split the rows into bands so that the grid still fills the chip
        const int64_t tiles0 = (a.N2 + 255) / 256;
        while (rb > TRB && tiles0 * ((a.N + rb - 1) / rb) * S < 512) rb = (rb / 2 + TRB - 1) / TRB * TRB;
    }
    a.RB = rb;
    const int64_t rblocks = (a.N + rb - 1) / rb;
    const int64_t tiles = (a.N2 + 255) / 256;
    // enough blocks to fill the chip (~16 per CU at five resident blocks each) while keeping the per-block row flush amortised
    static const int64_t gt_env = MXF_KNOB("MXF_BWD_GRID", 4096);
    int64_t ct = (tiles * rblocks * S + gt_env - 1) / gt_env;
    if (ct < 1) ct = 1;
    if (ct > 64) ct = 64;
    a.CT = (int)ct;
    dim3 g((unsigned)((tiles + ct - 1) / ct), (unsigned)rblocks, (unsigned)S);
    if (g.y > 65535u || g.z > 65535u) MXF_FAIL(h, -3, "mxf_gram_bwd: grid too large");
    const size_t shmem = (size_t)rb * QA * sizeof(T) + fixed;
    (void)FUSED;
    if (shmem > 64 * 1024)
        MXF_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_kernel<T, QT, KIND, PT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((gram_bwd_kernel<T, QT, KIND, PT>), g, dim3(256), shmem, st, a);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

template <typename T, int KIND, int PT>
int launch_bwd(mxf_ctx* h, const GramBwdArgs<T>& a, int S, hipStream_t st) {
    if (a.Q > 16) {
        if (PT > 0) MXF_FAIL(h, -3, "svgp fused reverse pass: Q > 16 runs through the generic (materialised dKuf) path");
        dim3 g((unsigned)((a.N2 + 255) / 256), (unsigned)(a.N < 65535 ? a.N : 65535), (unsigned)S);
        if (g.z > 65535u) MXF_FAIL(h, -3, "mxf_gram_bwd: grid too large");
        hipLaunchKernelGGL((gram_bwd_generic_kernel<T, KIND>), g, dim3(256), 0, st, a);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    if (a.Q <= 2) return launch_q<T, 2, KIND, PT>(h, a, S, st);
    if (a.Q <= 4) return launch_q<T, 4, KIND, PT>(h, a, S, st);
    if (a.Q <= 8) return launch_q<T, 8, KIND, PT>(h, a, S, st);
    return launch_q<T, 16, KIND, PT>(h, a, S, st);
}

template <typename T, int PT>
int launch_kind(mxf_ctx* h, int kind, const GramBwdArgs<T>& a, int S, hipStream_t st) {
    switch (kind) {
        case MXF_K_RBF: return launch_bwd<T, MXF_K_RBF, PT>(h, a, S, st);
        case MXF_K_MATERN12: return launch_bwd<T, MXF_K_MATERN12, PT>(h, a, S, st);
        case MXF_K_MATERN32: return launch_bwd<T, MXF_K_MATERN32, PT>(h, a, S, st);
        case MXF_K_MATERN52: return launch_bwd<T, MXF_K_MATERN52, PT>(h, a, S, st);
    }
    MXF_FAIL(h, -2, "mxf_gram_bwd: kind %d has no stationary reverse mode", kind);
}

template <typename T>
int bwd_typed(mxf_ctx* h, int kind, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t sX, const void* X2, int64_t sX2,
              const void* ls, int ard, int64_t sls, const void* var, int64_t svar, const void* dK, int64_t lddk, int64_t sdK,
              void* dX, void* dX2, void* dls, void* dvar, hipStream_t st, int dk_symmetric = 0) {
    GramBwdArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.square = (X2 == nullptr);
    a.sym = a.square ? dk_symmetric : 0;
    if (a.sym == 2 && Q > 16) MXF_FAIL(h, -3, "mxf_gram_bwd: the lower-triangle form needs Q <= 16");
    a.X = (const T*)X; a.X2 = a.square ? (const T*)X : (const T*)X2; a.sX = sX; a.sX2 = a.square ? sX : sX2;
    a.ls = (const T*)ls; a.sls = sls; a.var = (const T*)var; a.svar = svar; a.dK = (const T*)dK; a.lddk = lddk; a.sdK = sdK;
    a.dX = (T*)dX; a.dX2 = (T*)dX2; a.dls = (T*)dls; a.dvar = (T*)dvar;
    a.N = N; a.N2 = a.square ? N : N2; a.Q = Q; a.ard = ard;
    return launch_kind<T, 0>(h, kind, a, S, st);
}

// dst[i][0..7] = cs * src[i][0..Q-1] / l_q, zero padded; norms[i] = |dst[i]|^2 (optional).  Rows i = blockIdx.y * B + (row inside the
// sample), blockIdx.y = sample (the coordinates of the M inducing points: one "sample" of B = M rows).
// mx (optional; zeroed by the caller): bit pattern of max_i |aux[i]| (yv == nullptr: the row w) or of max_i |yv[s * sY + i % B] - aux[i]| (the
// residual y_n - U_n) -- the bound behind the f16 accumulation of the matrix-pipe pass.  One atomic per workgroup, and only if it can raise
// the word (non-negative floats order as their bit patterns).
// centre[q] = mean of Z[m][q] over the FIRST 64 inducing inputs (rows the caller appended to pad M -- far-away decoupled points, svgp_regression.py
// _pad_inducing -- come last and must not drag the centre away): the matrix-pipe pass forms r2 = |x|^2 + |z|^2 - 2 x.z in float32, whose absolute error grows with the NORMS of
// the scaled coordinates -- distances are translation invariant, so both operands are centred on the inducing inputs first (r04: inputs at an
// offset of 100 / 1000 units -- years, raw sensor readings -- gave 1e-2 / 98 % gradient errors and 1e-5 / 2e-2 on the bound un-centred)
__global__ __launch_bounds__(256) void bwd_centre_kernel(int64_t M, int Q, const float* __restrict__ Z, float* __restrict__ centre) {
    __shared__ double red[4];
    const int q = blockIdx.x;
    double s = 0.0;
    for (int64_t m = threadIdx.x; m < M; m += 256) s += (double)Z[m * Q + q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) centre[q] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)M);
}

__global__ __launch_bounds__(256) void bwd_prescale_kernel(const float* __restrict__ src, int64_t B, int Q, const float* __restrict__ ls, int ard,
                                                           const float* __restrict__ centre, float* __restrict__ dst, float* __restrict__ norms, float cs,
                                                           const float* __restrict__ aux, const float* __restrict__ yv, int64_t sY,
                                                           unsigned* __restrict__ mx) {
    __shared__ float smax[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = (int64_t)blockIdx.y * B + i;
    if (mx) {
        float m = 0.f;
        if (i < B) m = fabsf(yv ? yv[(int64_t)blockIdx.y * sY + i] - aux[r] : aux[r]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
            if (__builtin_bit_cast(unsigned, m) > *(volatile unsigned*)mx) atomicMax(mx, __builtin_bit_cast(unsigned, m));
        }
    }
    if (i >= B) return;
    float v[8], n2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = (q < Q) ? (src[r * Q + q] - centre[q]) / ls[ard ? q : 0] * cs : 0.f; n2 = fmaf(v[q], v[q], n2); }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<f32x4*>(dst + r * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + r * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    if (norms) norms[r] = n2;
}

int launch_mfma(mxf_ctx* h, int kind, int64_t M, int64_t SB, int64_t B, int Q, const float* Z, const float* X, const float* ls, int ard,
                const float* var, const float* Text, const float* Y, int64_t sY, const float* w, const float* noise, double a1, float* dZ,
                float* dX, float* dls, float* dvar, float* dY, int dY_shared, float* R, double* scal, hipStream_t st, int t_blocked,
                const unsigned* h0max, const unsigned* tmax) {
    const size_t nacc = ((size_t)M * 16 + 16) * sizeof(double);                                    // bytes, zeroed every call
    const size_t need = nacc + (((size_t)M + (size_t)SB) * 8 + (size_t)SB) * sizeof(float);        // + the scaled coordinates and |x_n|^2
    if (need > h->bwd_acc_bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) MXF_FAIL(h, -4, "svgp reverse pass: scratch must be allocated before a stream capture (run one eager step first)");
        if (h->bwd_acc) { (void)hipDeviceSynchronize(); (void)hipFree(h->bwd_acc); h->bwd_acc = nullptr; h->bwd_acc_bytes = 0; ++h->ws_generation; }
        if (hipMalloc((void**)&h->bwd_acc, need) != hipSuccess) { h->bwd_acc = nullptr; MXF_FAIL(h, -4, "svgp reverse pass: cannot allocate %zu bytes", need); }
        h->bwd_acc_bytes = need;
    }
    MXF_HIP(h, hipMemsetAsync(h->bwd_acc, 0, nacc, st));
    double* zacc = reinterpret_cast<double*>(h->bwd_acc);
    float* Zs = reinterpret_cast<float*>(reinterpret_cast<char*>(h->bwd_acc) + nacc);
    float* Xs = Zs + (size_t)M * 8;
    float* Xn = Xs + (size_t)SB * 8;
    const float cs = kind == MXF_K_RBF ? 0.84932180028801904272f : 1.f;      // RBF: exp(-r2 / 2) = 2^-(cs^2 r2), the bare v_exp_f32 in the pass
    // f16 accumulation (RBF; the T product's operand bound must be known: the split GEMM's max |H0| word)
    static const int f16_env = (int)MXF_KNOB("MXF_BWD_F16", 1);
    const bool f16 = f16_env && kind == MXF_K_RBF && h0max != nullptr;
    unsigned* mx = reinterpret_cast<unsigned*>(zacc + (size_t)M * 16 + 8);          // two words behind dls3[8], zeroed with the accumulators
    const float* Urow = Text + M * SB;
    if (SB > 2147483647LL - 4096) MXF_FAIL(h, -3, "svgp reverse pass: more than 2^31 columns (%lld)", (long long)SB);       // (32-bit column indices in the pass)
    const int64_t nsamp = SB / B;         // (the matrix-pipe pass requires B % 16 == 0 and whole samples: SB = S B)
    if (nsamp > 65535 || nsamp * B != SB) MXF_FAIL(h, -3, "svgp reverse pass: bad sample layout (SB %lld, B %lld)", (long long)SB, (long long)B);
    float* centre = reinterpret_cast<float*>(zacc + (size_t)M * 16 + 9);            // (eight floats behind the two bound words; written after the memset above)
    hipLaunchKernelGGL(bwd_centre_kernel, dim3((unsigned)Q), dim3(256), 0, st, M < 64 ? M : (int64_t)64, Q, Z, centre);
    hipLaunchKernelGGL(bwd_prescale_kernel, dim3((unsigned)((M + 255) / 256), 1), dim3(256), 0, st, Z, M, Q, ls, ard, (const float*)centre, Zs, (float*)nullptr, cs,
                       w, (const float*)nullptr, (int64_t)0, f16 ? mx : (unsigned*)nullptr);
    hipLaunchKernelGGL(bwd_prescale_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)nsamp), dim3(256), 0, st, X, B, Q, ls, ard, (const float*)centre, Xs, Xn, cs,
                       Urow, Y, sY, f16 ? mx + 1 : (unsigned*)nullptr);
    BwdMfmaArgs a;
    a.Zs = Zs; a.Xs = Xs; a.Xn = Xn; a.ls = ls; a.var = var; a.T = Text; a.U = Text + M * SB; a.Y = Y; a.w = w; a.noise = noise;
    a.dX = dX; a.dY = dY; a.zacc = zacc; a.dls3 = zacc + (size_t)M * 16; a.dvar = dvar; a.scal = scal;
    a.M = M; a.SB = SB; a.B = B; a.sY = sY; a.Q = Q; a.ard = ard; a.dY_shared = dY_shared; a.a1 = a1; a.tblk = t_blocked;
    a.h0max = h0max; a.mx = mx; a.tmax = tmax;
    const int64_t quads = (SB + 63) / 64, bands = (M + MF_RB - 1) / MF_RB;
    // work items: ~1024 (r03; was 8192).  Every workgroup ends with a flush of its row-side sums (LDS, then float64 atomics), a fixed cost
    // per workgroup: with 8192 of them the pass took 0.82 ms at 4 samples where 3.0 / 8 = 0.38 was its share (per-rank step 5.02 -> 4.63 ms
    // with 1024; 32 samples: 25.18 -> 24.79 ms; 512 measures the same, tests/probes/bwd_grid.sh)
    static const int64_t gt_env = MXF_KNOB("MXF_BWD_MFMA_GRID", 1024);
    int64_t ct = (quads * bands + gt_env - 1) / gt_env;
    if (ct < 1) ct = 1;
    if (ct > 256) ct = 256;
    a.CT = (int)ct;
    dim3 g((unsigned)((quads + ct - 1) / ct), (unsigned)bands, 1);
    if (g.y > 65535u) MXF_FAIL(h, -3, "svgp reverse pass: too many row bands");
    const bool full = (M % MF_RB == 0) && (SB % 64 == 0);
    // MXF_BWD_DMAT=1 (probe builds only): T tiles and column values through LDS-DMA (svgp_bwd_mfma_kernel DMAT) -- VERDICT r05 item 7's experiment.
    // Correct (tests/test_gpu_fullsize_oracle.py, test_gpu_sweep.py pass with it) and NOT faster: the pass alone 2.99-3.06 against 2.63-2.74 ms,
    // the step 22.1-22.5 either way (tests/probes/r06_bwd_dmat.sh).  With 56 instead of 32 KB per CU in flight nothing moves, i.e. the pass
    // does not wait for memory: at 2 waves per SIMD a row tile costs a SIMD ~630 cycles, about the sum of its ~340 VALU cycles (exp2, the
    // weights, two f16 splits) and its 8 MFMAs of 32 cycles -- the two do not overlap inside a wave's dependent chain, and the LDS read of
    // the tile is one more exposed latency per step.  What would help is fewer instructions per element (a 16 x 32 tile on
    // v_mfma_f32_16x16x32_f16), not deeper queues.
    static const int dmat_env = (int)MXF_KNOB("MXF_BWD_DMAT", 0);
    (void)dmat_env;
#define MF_GO(KIND)                                                                                             \
    do {                                                                                                        \
        if (full) hipLaunchKernelGGL((svgp_bwd_mfma_kernel<KIND, true, false>), g, dim3(256), 0, st, a);       \
        else hipLaunchKernelGGL((svgp_bwd_mfma_kernel<KIND, false, false>), g, dim3(256), 0, st, a);           \
    } while (0)
    switch (kind) {
        case MXF_K_RBF:
#ifdef MXF_PROBES
            if (f16 && full && dmat_env && t_blocked && MF_MT == 8) hipLaunchKernelGGL((svgp_bwd_mfma_kernel<MXF_K_RBF, true, true, true>), g, dim3(256), 0, st, a);
            else
#endif
            if (f16 && full) hipLaunchKernelGGL((svgp_bwd_mfma_kernel<MXF_K_RBF, true, true>), g, dim3(256), 0, st, a);
            else if (f16) hipLaunchKernelGGL((svgp_bwd_mfma_kernel<MXF_K_RBF, false, true>), g, dim3(256), 0, st, a);
            else MF_GO(MXF_K_RBF);
            break;
        case MXF_K_MATERN12: MF_GO(MXF_K_MATERN12); break;
        case MXF_K_MATERN32: MF_GO(MXF_K_MATERN32); break;
        case MXF_K_MATERN52: MF_GO(MXF_K_MATERN52); break;
        default: MXF_FAIL(h, -2, "svgp reverse pass: kind %d has no stationary reverse mode", kind);
    }
#undef MF_GO
    hipLaunchKernelGGL(svgp_bwd_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, M, Q, ard, (const float*)Zs, ls, (const double*)a.zacc, (const double*)a.dls3, dZ, dls, R,
                       (double)cs, var, kind == MXF_K_RBF ? dvar : (float*)nullptr);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

template <typename T>
int fused_typed(mxf_ctx* h, int kind, int64_t M, int64_t SB, int64_t B, int Q, int P, const void* Z, const void* Xall, const void* ls,
                int ard, const void* var, const void* Text, const void* Y, int64_t sY, const void* w, const void* noise, double a1,
                void* dZ, void* dXall, void* dls, void* dvar, void* dY, int dY_shared, void* R, double* scal, hipStream_t st, int t_blocked,
                const unsigned* h0max, const unsigned* tmax) {
    GramBwdArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.square = 0;
    a.X = (const T*)Z; a.X2 = (const T*)Xall; a.sX = 0; a.sX2 = 0;
    a.ls = (const T*)ls; a.var = (const T*)var; a.dK = (const T*)Text; a.lddk = SB;
    a.dX = (T*)dZ; a.dX2 = (T*)dXall; a.dls = (T*)dls; a.dvar = (T*)dvar;
    a.N = M; a.N2 = SB; a.Q = Q; a.ard = ard;
    a.U = (const T*)Text + M * SB; a.Y = (const T*)Y; a.sY = sY; a.B = B; a.w = (const T*)w; a.noise = (const T*)noise;
    a.dY = (T*)dY; a.dY_shared = dY_shared; a.R = (T*)R; a.scal = scal; a.a1 = a1; a.P = P;
    if constexpr (sizeof(T) == 4) {
        if (mxf_svgp_bwd_is_mfma(kind, MXF_F32, SB, B, Q, P, Text))
            return launch_mfma(h, kind, M, SB, B, Q, (const float*)Z, (const float*)Xall, (const float*)ls, ard, (const float*)var, (const float*)Text,
                               (const float*)Y, sY, (const float*)w, (const float*)noise, a1, (float*)dZ, (float*)dXall, (float*)dls, (float*)dvar,
                               (float*)dY, dY_shared, (float*)R, scal, st, t_blocked, h0max, tmax);
    }
    if (t_blocked && !mxf_svgp_bwd_reads_blocked(kind, sizeof(T) == 4 ? MXF_F32 : MXF_F64, SB, B, Q, P, Text))
        MXF_FAIL(h, -2, "svgp fused reverse pass: this shape reads T row-major");
    a.tblk = t_blocked;
    if (P == 1) return launch_kind<T, 1>(h, kind, a, 1, st);
    return launch_kind<T, PMAX_ALL>(h, kind, a, 1, st);
}

}  // namespace

int mxf_gram_bwd_internal(mxf_ctx* h, int kind, int dtype, int S, int64_t N, int64_t N2, int Q, const void* X, int64_t sX,
                          const void* X2, int64_t sX2, const void* ls, int ard, int64_t sls, const void* var, int64_t svar,
                          const void* dK, int64_t lddk, int64_t sdK, void* dX, void* dX2, void* dls, void* dvar, hipStream_t st, int dk_symmetric) {
    if (S <= 0 || N <= 0 || (X2 && N2 <= 0)) return 0;
    if (dtype == MXF_F32) return bwd_typed<float>(h, kind, S, N, N2, Q, X, sX, X2, sX2, ls, ard, sls, var, svar, dK, lddk, sdK, dX, dX2, dls, dvar, st, dk_symmetric);
    if (dtype == MXF_F64) return bwd_typed<double>(h, kind, S, N, N2, Q, X, sX, X2, sX2, ls, ard, sls, var, svar, dK, lddk, sdK, dX, dX2, dls, dvar, st, dk_symmetric);
    MXF_FAIL(h, -2, "mxf_gram_bwd: bad dtype %d", dtype);
}

// SVGP-fused reverse pass over Text = [H0; w^T] Kuf_all (rows 0..M-1: T, rows M..M+P-1: U); column-side output dXall is
// WRITTEN (not accumulated); dZ, dls, dvar, R, scal are accumulated into (caller zeroes); dY written or (shared) accumulated.
bool mxf_svgp_bwd_is_mfma(int kind, int dtype, int64_t SB, int64_t B, int Q, int P, const void* Text) {
    static const int mf_env = MXF_KNOB("MXF_BWD_MFMA", 1);
    // RBF only (r04): the matrix-pipe pass forms r2 = |x|^2 + |z|^2 - 2 x.z in float32 -- absolute error ~1e-7 (|x|^2 + |z|^2).  The RBF weight is
    // smooth in r2; the Matern slopes are not (dk/dr2 = -k / 2r for Matern12): with inducing inputs next to data points -- Z = X[:M] is the
    // usual initialisation -- its dX / dZ came out 10-20 % off for Matern12 and 1e-3 off for Matern32 / 52 at Q = 3 ... 8, against 1e-6 ... 5e-5 from the
    // difference-form pass (tests/probes/bwd_form_accuracy.py).  MXF_BWD_MFMA=2 (probe build) puts the Matern kinds back on it.
    if (kind != MXF_K_RBF && mf_env != 2) return false;
    return mf_env && dtype == MXF_F32 && P == 1 && Q <= 8 && SB % 4 == 0 && SB >= 16 && B % 16 == 0 && ((uintptr_t)Text % 16) == 0;
}

// may T be handed over in 16-column blocks?  The matrix-pipe pass (RBF) requires it; the difference-form pass (Matern kinds, P > 1, Q > 8) reads
// either layout for Q <= 16 -- so that those calls keep the wide blocked-output T product (11.3 instead of 13.4 ms at the bench shape)
bool mxf_svgp_bwd_reads_blocked(int kind, int dtype, int64_t SB, int64_t B, int Q, int P, const void* Text) {
    return mxf_svgp_bwd_is_mfma(kind, dtype, SB, B, Q, P, Text) || (dtype == MXF_F32 && Q <= 16 && P <= PMAX_ALL && SB % 16 == 0);
}

// ---- r05: the reverse pass as the epilogue of the T product (mxf_fuse_args, internal.h; gemm_split.hip wide_body<..., FUSE>) ----------------------
// dY = -c1 e and sum_n e_n^2 per sample (the separate pass did this on its first row band)
__global__ __launch_bounds__(256) void fuse_resid_kernel(int64_t SB, int64_t B, const float* __restrict__ U, const float* __restrict__ Y, int64_t sY,
                                                         const float* __restrict__ noise, double a1, float* __restrict__ dY, int dY_shared,
                                                         double* __restrict__ scal) {
    __shared__ double red[16];
    const int64_t smp = blockIdx.y;
    const float c1 = (float)a1 / noise[0];
    double es = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        const int64_t n = smp * B + i;
        const float e = Y[smp * sY + i] - U[n];
        es += (double)e * (double)e;
        if (dY) { const float gy = -c1 * e; if (dY_shared) atomic_add(dY + i, gy); else dY[n] = gy; }
    }
    es = block_sum<double>(es, red);
    if (threadIdx.x == 0) atomic_add(scal + 2 * smp + 1, es);
}

int mxf_svgp_bwd_fuse_ok(int kind, int dtype, int64_t M, int64_t SB, int64_t B, int Q, int P) {
    static const int env = (int)MXF_KNOB("MXF_SVGP_FUSE", 0);
    return env && kind == MXF_K_RBF && dtype == MXF_F32 && P == 1 && Q <= 8 && (M % 256) == 0 && M >= 256 && (B % 256) == 0 && (SB % B) == 0 &&
           SB / B <= 65535 && SB < 2147483647LL - 4096;
}

int mxf_svgp_bwd_fuse_prepare(mxf_ctx* h, int64_t M, int64_t SB, int64_t B, int Q, const float* Z, const float* X, const float* ls, int ard,
                              const float* var, const float* U, const float* Y, int64_t sY, const float* w, const float* noise, double a1, float* dX,
                              float* dY, int dY_shared, double* scal, const unsigned* h0max, mxf_fuse_args* out, hipStream_t st) {
    const size_t nacc = ((size_t)M * 16 + 16) * sizeof(double);
    const size_t need = nacc + (((size_t)M + (size_t)SB) * 8 + (size_t)SB + (size_t)M) * sizeof(float);      // + scaled coordinates, |x_n|^2, |z_m|^2
    if (need > h->bwd_acc_bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) MXF_FAIL(h, -4, "svgp reverse pass: scratch must be allocated before a stream capture (run one eager step first)");
        if (h->bwd_acc) { (void)hipDeviceSynchronize(); (void)hipFree(h->bwd_acc); h->bwd_acc = nullptr; h->bwd_acc_bytes = 0; ++h->ws_generation; }
        if (hipMalloc((void**)&h->bwd_acc, need) != hipSuccess) { h->bwd_acc = nullptr; MXF_FAIL(h, -4, "svgp reverse pass: cannot allocate %zu bytes", need); }
        h->bwd_acc_bytes = need;
    }
    MXF_HIP(h, hipMemsetAsync(h->bwd_acc, 0, nacc, st));
    double* zacc = reinterpret_cast<double*>(h->bwd_acc);
    float* Zs = reinterpret_cast<float*>(reinterpret_cast<char*>(h->bwd_acc) + nacc);
    float* Xs = Zs + (size_t)M * 8;
    float* Xn = Xs + (size_t)SB * 8;
    float* Zn = Xn + (size_t)SB;
    const float cs = 0.84932180028801904272f;
    unsigned* mx = reinterpret_cast<unsigned*>(zacc + (size_t)M * 16 + 8);
    float* centre = reinterpret_cast<float*>(zacc + (size_t)M * 16 + 9);
    const int64_t nsamp = SB / B;
    hipLaunchKernelGGL(bwd_centre_kernel, dim3((unsigned)Q), dim3(256), 0, st, M < 64 ? M : (int64_t)64, Q, Z, centre);
    hipLaunchKernelGGL(bwd_prescale_kernel, dim3((unsigned)((M + 255) / 256), 1), dim3(256), 0, st, Z, M, Q, ls, ard, (const float*)centre, Zs, Zn, cs,
                       w, (const float*)nullptr, (int64_t)0, mx);
    hipLaunchKernelGGL(bwd_prescale_kernel, dim3((unsigned)((B + 255) / 256), (unsigned)nsamp), dim3(256), 0, st, X, B, Q, ls, ard, (const float*)centre, Xs, Xn, cs,
                       U, Y, sY, mx + 1);
    const int64_t gx = (B + 255) / 256 > 64 ? 64 : (B + 255) / 256;
    hipLaunchKernelGGL(fuse_resid_kernel, dim3((unsigned)gx, (unsigned)nsamp), dim3(256), 0, st, SB, B, U, Y, sY, noise, a1, dY, dY_shared, scal);
    MXF_LAUNCH_CHECK(h);
    out->Zs = Zs; out->Zn = Zn; out->Xs = Xs; out->Xn = Xn; out->U = U; out->Y = Y; out->w = w; out->ls = ls; out->var = var; out->noise = noise;
    out->dX = dX; out->zacc = zacc; out->dls3 = zacc + (size_t)M * 16; out->scal = scal; out->h0max = h0max; out->mx = mx;
    out->B = B; out->sY = sY; out->Q = Q; out->ard = ard; out->a1 = a1;
    return 0;
}

int mxf_svgp_bwd_fuse_finish(mxf_ctx* h, int64_t M, int Q, int ard, const float* ls, const float* var, const mxf_fuse_args* fz, float* dZ, float* dls,
                             float* dvar, float* R, hipStream_t st) {
    hipLaunchKernelGGL(svgp_bwd_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, M, Q, ard, fz->Zs, ls, (const double*)fz->zacc,
                       (const double*)fz->dls3, dZ, dls, R, (double)0.84932180028801904272f, var, dvar);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

int mxf_svgp_bwd_fused_internal(mxf_ctx* h, int kind, int dtype, int64_t M, int64_t SB, int64_t B, int Q, int P, const void* Z,
                                const void* Xall, const void* ls, int ard, const void* var, const void* Text, const void* Y,
                                int64_t sY, const void* w, const void* noise, double a1, void* dZ, void* dXall, void* dls,
                                void* dvar, void* dY, int dY_shared, void* R, double* scal, hipStream_t st, int t_blocked,
                                const unsigned* h0max, const unsigned* tmax) {
    if (P > PMAX_ALL) MXF_FAIL(h, -3, "svgp fused reverse pass: P > %d", PMAX_ALL);
    if (dtype == MXF_F32) return fused_typed<float>(h, kind, M, SB, B, Q, P, Z, Xall, ls, ard, var, Text, Y, sY, w, noise, a1, dZ, dXall, dls, dvar, dY, dY_shared, R, scal, st, t_blocked, h0max, tmax);
    if (dtype == MXF_F64) return fused_typed<double>(h, kind, M, SB, B, Q, P, Z, Xall, ls, ard, var, Text, Y, sY, w, noise, a1, dZ, dXall, dls, dvar, dY, dY_shared, R, scal, st, t_blocked, h0max, tmax);
    MXF_FAIL(h, -2, "svgp fused reverse pass: bad dtype %d", dtype);
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

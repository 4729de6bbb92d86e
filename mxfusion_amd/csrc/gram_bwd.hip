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

    T il[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) il[q] = (q < Q) ? (T)1 / ls[a.ard ? q : 0] : (T)0;
    for (int64_t i = tid; i < (rend - r0) * QA; i += 256) racc[i] = (T)0;

    T gl[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) gl[q] = 0;
    T gvar = 0;
    const T beta = FUSED ? (T)1 / a.noise[0] : (T)0;
    const T c1 = FUSED ? (T)a.a1 * beta : (T)0;

    for (int ct = 0; ct < a.CT; ++ct) {
        const int64_t tile0 = ((int64_t)blockIdx.x * a.CT + ct) * 256;
        if (tile0 >= a.N2) break;
        const int64_t col = tile0 + tid;
        const bool cvalid = col < a.N2;
        T z[QT], gz[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) { z[q] = (cvalid && q < Q) ? X2[col * Q + q] * il[q] : (T)0; gz[q] = 0; }
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
                xs[i] = (row < rend && q < Q) ? X[row * Q + q] / ls[a.ard ? q : 0] : (T)0;
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
                  pre[u] = (cvalid && r0_ + u < rmax) ? dK[(rt + r0_ + u) * a.lddk + col] : (T)0;
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
                if (a.dX) rs = row_reduce_scatter<T, QT>(tq, lane);
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
                    const bool act = isq ? (a.dX != nullptr && l16 < Q) : (a.R != nullptr && l16 < QT + PMAX && rp < P);
                    if (lane < 16 && act) lds_add(ra + (isq ? l16 : QT + rp), v);
                } else if constexpr (sizeof(T) == 4) {
                    if (a.dX) { const float v = wave_rows_sum(rs); if (lane < QT && lane < Q) lds_add(ra + lane, v); }
                    if (FUSED && a.R) { const float v = wave_rows_sum(rsR); if (lane < PMAX && lane < P) lds_add(ra + QT + lane, v); }
                } else {
                    if (a.dX && l16 < QT && l16 < Q) lds_add(ra + l16, rs);
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
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if (q < Q) {
                    T* p = dXc + (int64_t)s * sXc + col * Q + q;
                    if (plain) *p += gz[q] * il[q]; else atomic_add(p, gz[q] * il[q]);
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
    if (a.dX || (FUSED && a.R)) {
        for (int64_t i = tid; i < (rend - r0) * QA; i += 256) {
            const int64_t r = i / QA;
            const int c = (int)(i % QA);
            if (c < QT) {
                if (a.dX && c < Q) atomic_add(a.dX + (int64_t)s * a.sX + (r0 + r) * Q + c, racc[i] / ls[a.ard ? c : 0]);
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

template <typename T, int QT, int KIND, int PT>
int launch_q(mxf_ctx* h, GramBwdArgs<T> a, int S, hipStream_t st) {
    constexpr bool FUSED = PT > 0;
    constexpr int QA = QT + PT;
    const size_t fixed = (size_t)(TRB * QT + TRB * PT + 16) * sizeof(T) + 16 * sizeof(double) + 64;
    // LDS per block (the row accumulators): 30 KB = five blocks per CU; 80 KB (two blocks, fewer row flushes) measured 1 % slower per step
    static const int bud_env = getenv("MXF_BWD_LDS_KB") ? atoi(getenv("MXF_BWD_LDS_KB")) : 30;
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
    static const int64_t gt_env = getenv("MXF_BWD_GRID") ? atoll(getenv("MXF_BWD_GRID")) : 4096;
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
              void* dX, void* dX2, void* dls, void* dvar, hipStream_t st) {
    GramBwdArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.square = (X2 == nullptr);
    a.X = (const T*)X; a.X2 = a.square ? (const T*)X : (const T*)X2; a.sX = sX; a.sX2 = a.square ? sX : sX2;
    a.ls = (const T*)ls; a.sls = sls; a.var = (const T*)var; a.svar = svar; a.dK = (const T*)dK; a.lddk = lddk; a.sdK = sdK;
    a.dX = (T*)dX; a.dX2 = (T*)dX2; a.dls = (T*)dls; a.dvar = (T*)dvar;
    a.N = N; a.N2 = a.square ? N : N2; a.Q = Q; a.ard = ard;
    return launch_kind<T, 0>(h, kind, a, S, st);
}

template <typename T>
int fused_typed(mxf_ctx* h, int kind, int64_t M, int64_t SB, int64_t B, int Q, int P, const void* Z, const void* Xall, const void* ls,
                int ard, const void* var, const void* Text, const void* Y, int64_t sY, const void* w, const void* noise, double a1,
                void* dZ, void* dXall, void* dls, void* dvar, void* dY, int dY_shared, void* R, double* scal, hipStream_t st) {
    GramBwdArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.square = 0;
    a.X = (const T*)Z; a.X2 = (const T*)Xall; a.sX = 0; a.sX2 = 0;
    a.ls = (const T*)ls; a.var = (const T*)var; a.dK = (const T*)Text; a.lddk = SB;
    a.dX = (T*)dZ; a.dX2 = (T*)dXall; a.dls = (T*)dls; a.dvar = (T*)dvar;
    a.N = M; a.N2 = SB; a.Q = Q; a.ard = ard;
    a.U = (const T*)Text + M * SB; a.Y = (const T*)Y; a.sY = sY; a.B = B; a.w = (const T*)w; a.noise = (const T*)noise;
    a.dY = (T*)dY; a.dY_shared = dY_shared; a.R = (T*)R; a.scal = scal; a.a1 = a1; a.P = P;
    if (P == 1) return launch_kind<T, 1>(h, kind, a, 1, st);
    return launch_kind<T, PMAX_ALL>(h, kind, a, 1, st);
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

// SVGP-fused reverse pass over Text = [H0; w^T] Kuf_all (rows 0..M-1: T, rows M..M+P-1: U); column-side output dXall is
// WRITTEN (not accumulated); dZ, dls, dvar, R, scal are accumulated into (caller zeroes); dY written or (shared) accumulated.
int mxf_svgp_bwd_fused_internal(mxf_ctx* h, int kind, int dtype, int64_t M, int64_t SB, int64_t B, int Q, int P, const void* Z,
                                const void* Xall, const void* ls, int ard, const void* var, const void* Text, const void* Y,
                                int64_t sY, const void* w, const void* noise, double a1, void* dZ, void* dXall, void* dls,
                                void* dvar, void* dY, int dY_shared, void* R, double* scal, hipStream_t st) {
    if (P > PMAX_ALL) MXF_FAIL(h, -3, "svgp fused reverse pass: P > %d", PMAX_ALL);
    if (dtype == MXF_F32) return fused_typed<float>(h, kind, M, SB, B, Q, P, Z, Xall, ls, ard, var, Text, Y, sY, w, noise, a1, dZ, dXall, dls, dvar, dY, dY_shared, R, scal, st);
    if (dtype == MXF_F64) return fused_typed<double>(h, kind, M, SB, B, Q, P, Z, Xall, ls, ard, var, Text, Y, sY, w, noise, a1, dZ, dXall, dls, dvar, dY, dY_shared, R, scal, st);
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

// Batched MFMA GEMM for gfx950:  C = alpha * op(A) * op(B) + beta * C   (row-major, f32 / f64).
//
// Replaces MXNet linalg.gemm2 / linalg.syrk call sites of the reference and is the
// building block of the blocked Cholesky / triangular solves and of the SVGP contractions.
//
// Roofline: MFMA bound.  f32 uses v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = 157 TF chip
// peak), f64 uses v_mfma_f64_16x16x4_f64.  At 1/16 of the bf16 MFMA rate the matrix pipe is the only
// thing that matters: a 128x128x16 block tile (4 waves as 2x2, 64x64 per wave) needs just 32 LDS
// fragment reads per 2048+ MFMA cycles, so operands are staged with plain guarded loads (any
// alignment / ragged edge) into k-major LDS tiles and read with conflict-free ds_read_b32/b64.
//   A-fragment (32x32x2): lane l -> A[i = l&31][k = l>>5];  B-fragment: B[k = l>>5][j = l&31]
//   A-fragment (16x16x4 f64): lane l -> A[i = l&15][k = l>>4]; B[k = l>>4][j = l&15]
//   C/D f32 32x32: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
//   C/D f64 16x16: col = l&15, row = (l>>4) + 4*r, r in [0,4)
#include "common.h"
#include "internal.h"
#include <stdlib.h>

namespace {

#ifndef MXF_GEMM_BK
#define MXF_GEMM_BK 32
#endif
#ifndef MXF_GEMM_WPS
#define MXF_GEMM_WPS 2
#endif
#ifndef MXF_GEMM_WAVES
#define MXF_GEMM_WAVES 8
#endif
constexpr int BM = 128, BN = 128, BK = MXF_GEMM_BK;
constexpr int NWAVE = MXF_GEMM_WAVES;          // 4: 2x2 waves of 64x64;  8: 2x4 waves of 64x32 (half the accumulators -> 4 waves/SIMD)
constexpr int NT = 64 * NWAVE;
constexpr int WN = (NWAVE == 8) ? 32 : 64;     // wave tile width

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Tile;
template <> struct Tile<float> { static constexpr int LD = BM + 4; };    // row stride of the k-major LDS tiles (16-B aligned rows)
template <> struct Tile<double> { static constexpr int LD = BM + 2; };

template <typename T>
struct GemmArgs {
    const T* A; const T* B; T* C;
    int64_t M, N, K, lda, ldb, ldc, sA, sB, sC;
    T alpha, beta;
    int splitk, lower_only, atomic, vecA, vecB;
    int k_from_m;     // triangular op(A): 1 = op(A) upper triangular (A^T of a lower-triangular A): tile rows m0.. only need k >= m0;
                      //                 2 = op(A) lower triangular (A itself, not transposed): they only need k < m0 + tile rows
    int64_t kchunk;
    int64_t tm, tn, ntiles, nwg;   // tile grid, tiles per (batch,split), total workgroups
    int rev_m;
    int64_t xc_max;                // > 0: balanced triangular mapping (lower_only with k_from_m == 1), 8 * xc_max workgroups per batch entry
};

constexpr int EPT = BK * 128 / NT;    // elements of one operand tile per thread

// ---- generic (guarded, any alignment) tile loaders: [BK x 128] tile, EPT scalars per thread ------------------
// KCONT: the matrix is stored with k contiguous (A not transposed / B transposed)
template <typename T, bool KCONT>
__device__ __forceinline__ void load_tile(T (&reg)[EPT], const T* __restrict__ P, int64_t ld, int64_t mn0, int64_t MN,
                                          int64_t k0, int64_t kend, int tid) {
    if (KCONT) {
        const int k = tid & (BK - 1);
        const int64_t kk = k0 + k;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int64_t i = mn0 + (tid / BK) + (NT / BK) * j;
            reg[j] = (i < MN && kk < kend) ? P[i * ld + kk] : (T)0;
        }
    } else {
        const int i = tid & 127;
        const int64_t ii = mn0 + i;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int64_t kk = k0 + (tid >> 7) + (NT / 128) * j;
            reg[j] = (ii < MN && kk < kend) ? P[kk * ld + ii] : (T)0;
        }
    }
}

template <typename T, bool KCONT>
__device__ __forceinline__ void store_tile(const T (&reg)[EPT], T* __restrict__ S, int tid) {
    constexpr int LD = Tile<T>::LD;
    if (KCONT) {
        const int k = tid & (BK - 1);
#pragma unroll
        for (int j = 0; j < EPT; ++j) S[k * LD + (tid / BK) + (NT / BK) * j] = reg[j];
    } else {
        const int i = tid & 127;
#pragma unroll
        for (int j = 0; j < EPT; ++j) S[((tid >> 7) + (NT / 128) * j) * LD + i] = reg[j];
    }
}

// ---- fast (interior, 16-byte aligned) tile loaders: EPT/VEC 16-byte vectors per thread ----------------------------
template <typename T, bool KCONT>
__device__ __forceinline__ void load_tile_vec(T (&reg)[EPT], const T* __restrict__ P, int64_t ld, int64_t mn0, int64_t k0, int tid) {
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    constexpr int NV = EPT / VEC;
    if (KCONT) {
        constexpr int VPR = BK / VEC;                 // vectors per row (k direction)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + NT * j, i = v / VPR, kc = v % VPR;
            *reinterpret_cast<V*>(&reg[j * VEC]) = *reinterpret_cast<const V*>(P + (mn0 + i) * ld + k0 + kc * VEC);
        }
    } else {
        constexpr int VPR = 128 / VEC;                // vectors per k-row (mn direction)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + NT * j, kr = v / VPR, c = v % VPR;
            *reinterpret_cast<V*>(&reg[j * VEC]) = *reinterpret_cast<const V*>(P + (k0 + kr) * ld + mn0 + c * VEC);
        }
    }
}

template <typename T, bool KCONT>
__device__ __forceinline__ void store_tile_vec(const T (&reg)[EPT], T* __restrict__ S, int tid) {
    constexpr int LD = Tile<T>::LD;
    constexpr int VEC = Vec16<T>::n;
    typedef typename Vec16<T>::type V;
    constexpr int NV = EPT / VEC;
    if (KCONT) {
        constexpr int VPR = BK / VEC;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + NT * j, i = v / VPR, kc = v % VPR;
#pragma unroll
            for (int e = 0; e < VEC; ++e) S[(kc * VEC + e) * LD + i] = reg[j * VEC + e];
        }
    } else {
        constexpr int VPR = 128 / VEC;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + NT * j, kr = v / VPR, c = v % VPR;
            *reinterpret_cast<V*>(S + kr * LD + c * VEC) = *reinterpret_cast<const V*>(&reg[j * VEC]);
        }
    }
}

template <typename T> struct Acc;
template <> struct Acc<float> {
    static constexpr int NB_ = WN / 32;      // 32-wide MFMA tiles along n per wave
    f32x16 c[2][NB_];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB_; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[a][b][r] = 0.f;
    }
    // one BK slab from LDS
    __device__ __forceinline__ void mma(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int lane) {
        constexpr int LD = Tile<float>::LD;
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            float a[2], b[NB_];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = As[(ks + lk) * LD + wm + 32 * t + li];
#pragma unroll
            for (int t = 0; t < NB_; ++t) b[t] = Bs[(ks + lk) * LD + wn + 32 * t + li];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < NB_; ++y) c[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], c[x][y], 0, 0, 0);
        }
    }
    template <typename F> __device__ __forceinline__ void for_each(int wm, int wn, int lane, F f) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB_; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int col = wn + b * 32 + (lane & 31);
                    f(row, col, c[a][b][r]);
                }
    }
};
template <> struct Acc<double> {
    static constexpr int NB_ = WN / 16;      // 16-wide MFMA tiles along n per wave
    f64x4 c[4][NB_];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NB_; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) c[a][b][r] = 0.0;
    }
    __device__ __forceinline__ void mma(const double* __restrict__ As, const double* __restrict__ Bs, int wm, int wn, int lane) {
        constexpr int LD = Tile<double>::LD;
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            double a[4], b[NB_];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = As[(ks + lk) * LD + wm + 16 * t + li];
#pragma unroll
            for (int t = 0; t < NB_; ++t) b[t] = Bs[(ks + lk) * LD + wn + 16 * t + li];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < NB_; ++y) c[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], c[x][y], 0, 0, 0);
        }
    }
    template <typename F> __device__ __forceinline__ void for_each(int wm, int wn, int lane, F f) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NB_; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm + a * 16 + (lane >> 4) + 4 * r;
                    const int col = wn + b * 16 + (lane & 15);
                    f(row, col, c[a][b][r]);
                }
    }
};

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(NT, (NWAVE == 8 ? (sizeof(T) == 4 ? 4 : 2) : (sizeof(T) == 4 ? MXF_GEMM_WPS : 1))) void gemm_kernel(GemmArgs<T> g) {
    constexpr int LD = Tile<T>::LD;
    __shared__ __attribute__((aligned(16))) T smem[2][2][BK * LD];   // [buffer][A|B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (NWAVE == 8) ? (wave >> 2) * 64 : (wave >> 1) * 64;
    const int wn = (NWAVE == 8) ? (wave & 3) * 32 : (wave & 1) * 64;
    // XCD-aware 1-D work mapping.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness): give
    // each XCD a CONTIGUOUS range of work ids so that tiles sharing an operand panel hit the same private L2, and
    // enumerate only the tiles that exist (lower_only: compact triangular decode) so every XCD gets the same load.
    int64_t wid = blockIdx.x;
    if (!g.xc_max) {
        const int64_t q = g.nwg / 8, r = g.nwg % 8, xcd = wid % 8, j = wid / 8;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int64_t zs = wid / g.ntiles;                  // (batch, split) slowest
    int64_t t = wid % g.ntiles;
    int64_t tile_m, tile_n;
    if (g.xc_max) {
        // Lower tiles of a product whose tile row r only needs k >= r BM (K^-1 = L^-T L^-1): row r has r + 1 tiles of (tm - r) k blocks
        // each, so contiguous tile ranges per XCD are badly unbalanced (the first eighth of the tiles carries 2.4x its share at tm = 64).
        // XCD x takes the rows r = x (mod 8) -- equal work to 1 % -- longest k loops first; the tile counts differ per XCD, so the grid
        // holds 8 * xc_max workgroups per batch entry and the surplus ones leave.
        const int64_t per = 8 * g.xc_max, rem = (int64_t)blockIdx.x % per;
        zs = (int64_t)blockIdx.x / per;
        int64_t row = rem & 7, j = rem >> 3;
        while (row < g.tm && j >= row + 1) { j -= row + 1; row += 8; }
        if (row >= g.tm) return;
        tile_m = row; tile_n = j;
    } else if (g.lower_only) {                    // t -> (row, col), col <= row
        int64_t row = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        tile_m = row; tile_n = t - row * (row + 1) / 2;
    } else {                                      // row tiles fastest: consecutive ids share the B (column) panel
        tile_m = t % g.tm; tile_n = t / g.tm;
        if (g.k_from_m == 2 && g.rev_m) tile_m = g.tm - 1 - tile_m;      // k < m0 + BM: the long k loops first
    }
    const int64_t m0 = tile_m * BM, n0 = tile_n * BN;
    const int batch = (int)(zs / g.splitk), split = (int)(zs % g.splitk);

    const T* __restrict__ A = g.A + (int64_t)batch * g.sA;
    const T* __restrict__ B = g.B + (int64_t)batch * g.sB;
    T* __restrict__ C = g.C + (int64_t)batch * g.sC;

    int64_t kbeg = (int64_t)split * g.kchunk;
    int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    if (g.k_from_m == 1) { const int64_t kf = m0 / BK * BK; if (kf > kbeg) kbeg = kf; }
    if (g.k_from_m == 2) { const int64_t kl = m0 + BM; if (kl < kend) kend = kl; }

    Acc<T> acc;
    acc.zero();

    T ra[EPT], rb[EPT];
    // interior, 16-byte-aligned tiles take the vector loaders (block-uniform choice)
    const bool fullk = ((kend - kbeg) % BK) == 0;
    const bool fastA = g.vecA && fullk && (m0 + BM <= g.M);
    const bool fastB = g.vecB && fullk && (n0 + BN <= g.N);
#define LOAD_A(k0) do { if (fastA) load_tile_vec<T, !TA>(ra, A, g.lda, m0, (k0), tid); else load_tile<T, !TA>(ra, A, g.lda, m0, g.M, (k0), kend, tid); } while (0)
#define LOAD_B(k0) do { if (fastB) load_tile_vec<T, TB>(rb, B, g.ldb, n0, (k0), tid); else load_tile<T, TB>(rb, B, g.ldb, n0, g.N, (k0), kend, tid); } while (0)
#define STORE_A(buf) do { if (fastA) store_tile_vec<T, !TA>(ra, smem[buf][0], tid); else store_tile<T, !TA>(ra, smem[buf][0], tid); } while (0)
#define STORE_B(buf) do { if (fastB) store_tile_vec<T, TB>(rb, smem[buf][1], tid); else store_tile<T, TB>(rb, smem[buf][1], tid); } while (0)
    if (kbeg < kend) {
        LOAD_A(kbeg); LOAD_B(kbeg);
        STORE_A(0); STORE_B(0);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK < kend);
        if (more) { LOAD_A(k0 + BK); LOAD_B(k0 + BK); }
        acc.mma(smem[cur][0], smem[cur][1], wm, wn, lane);
        if (more) { STORE_A(cur ^ 1); STORE_B(cur ^ 1); }
        __syncthreads();
        cur ^= 1;
    }
#undef LOAD_A
#undef LOAD_B
#undef STORE_A
#undef STORE_B

    const T alpha = g.alpha, beta = g.beta;
    const bool atomic = g.atomic != 0;
    acc.for_each(wm, wn, lane, [&](int r, int c, T v) {
        const int64_t row = m0 + r, col = n0 + c;
        if (row < g.M && col < g.N && !(g.lower_only && col > row)) {
            T* p = C + row * g.ldc + col;
            if (atomic) atomic_add(p, alpha * v);
            else *p = (beta == (T)0) ? alpha * v : alpha * v + beta * (*p);
        }
    });
}

// ---- float64 C = alpha A B^T + beta C (A: M x K, B: N x K, both with k contiguous), interior tiles only: LDS-DMA, three buffers ----------
// The generic kernel stages every operand block through registers (global -> VGPR -> ds_write with a transposition -> barrier); with one
// 133 KB workgroup per CU nothing covers that phase and the matrix pipe stands at 57-61 %.  Here the blocks go
// global -> LDS by DMA (global_load_lds_dwordx4, no staging registers, no store phase), two blocks ahead into a three-slot ring, and stay
// in their [row][k] form: a 16-wide k block of a row is 128 bytes = eight 16-byte units, unit u of row r stored at position u ^ ((r >> 1) & 7)
// (the swizzle is applied to the SOURCE address, the DMA writes lane-contiguously), which makes the MFMA operand reads -- lane (li, lk)
// reads element (row li, k = ks + lk) -- conflict-free: 16 rows x 2 halves of a unit cover the 64 banks exactly once per half wave.
// Tile 128 x 128, BK = 16, eight waves of 64 x 32 (as the generic kernel), 96 KB of LDS.  Requirements (checked by the launcher):
// M, N multiples of 128, every k range a multiple of 16, 16-byte aligned rows.
// An operand stored the other way round (K x M with the rows of the tile contiguous: A of a transposed-A product, B of a plain-B one) comes
// in as its 16 k rows of 1 KB, one DMA instruction each, at an LDS row stride of 144 doubles (1 KB + 128 B: the two k values a half wave
// reads fall on the two halves of the banks).  AKM / BKM: operand stored as (k, m).
constexpr int DBK = 16, DKM_LD = 144;
template <bool AKM, bool BKM>
__global__ __launch_bounds__(512, 2) void gemm_f64_dma_kernel(GemmArgs<double> g) {
    constexpr int ASZ = AKM ? DBK * DKM_LD : 128 * DBK, BSZ = BKM ? DBK * DKM_LD : 128 * DBK;
    __shared__ __attribute__((aligned(16))) double smem[3][ASZ + BSZ];      // [slot][A slab | B slab]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;
    int64_t wid = blockIdx.x;
    if (!g.xc_max) {
        const int64_t q = g.nwg / 8, r = g.nwg % 8, xcd = wid % 8, j = wid / 8;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int64_t zs = wid / g.ntiles;
    int64_t t = wid % g.ntiles;
    int64_t tile_m, tile_n;
    if (g.xc_max) {                              // balanced triangular mapping, as gemm_kernel
        const int64_t per = 8 * g.xc_max, rem = (int64_t)blockIdx.x % per;
        zs = (int64_t)blockIdx.x / per;
        int64_t row = rem & 7, j = rem >> 3;
        while (row < g.tm && j >= row + 1) { j -= row + 1; row += 8; }
        if (row >= g.tm) return;
        tile_m = row; tile_n = j;
    } else if (g.lower_only) {
        int64_t row = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        tile_m = row; tile_n = t - row * (row + 1) / 2;
    } else {
        tile_m = t % g.tm; tile_n = t / g.tm;
        if (g.k_from_m == 2 && g.rev_m) tile_m = g.tm - 1 - tile_m;
    }
    const int64_t m0 = tile_m * BM, n0 = tile_n * BN;
    const int batch = (int)(zs / g.splitk), split = (int)(zs % g.splitk);
    const double* __restrict__ A = g.A + (int64_t)batch * g.sA;
    const double* __restrict__ B = g.B + (int64_t)batch * g.sB;
    double* __restrict__ C = g.C + (int64_t)batch * g.sC;
    int64_t kbeg = (int64_t)split * g.kchunk;
    int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    if (g.k_from_m == 1) { const int64_t kf = m0 / BK * BK; if (kf > kbeg) kbeg = kf; }
    if (g.k_from_m == 2) { const int64_t kl = m0 + BM; if (kl < kend) kend = kl; }
    const int64_t nk = kend > kbeg ? (kend - kbeg) / DBK : 0;

    Acc<double> acc;
    acc.zero();
    // DMA sources of this lane, two instructions per operand and block.  (m, k) storage: instruction j covers rows 16 w + 8 j .. + 8, lane l
    // -> row (l >> 3), position (l & 7), fetching unit pos ^ ((row >> 1) & 7).  (k, m) storage: instruction j is k row 2 w + j, lane l -> m = 2 l.
    const double* srcA[2];
    const double* srcB[2];
    int dstA[2], dstB[2];
    int64_t stepA, stepB;                 // source advance per k block
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (AKM) { const int kr = 2 * wave + j; srcA[j] = A + (kbeg + kr) * g.lda + m0 + 2 * lane; dstA[j] = kr * DKM_LD; }
        else { const int row = 16 * wave + 8 * j + (lane >> 3); srcA[j] = A + (m0 + row) * g.lda + kbeg + 2 * ((lane & 7) ^ ((row >> 1) & 7)); dstA[j] = (16 * wave + 8 * j) * DBK; }
        if (BKM) { const int kr = 2 * wave + j; srcB[j] = B + (kbeg + kr) * g.ldb + n0 + 2 * lane; dstB[j] = ASZ + kr * DKM_LD; }
        else { const int row = 16 * wave + 8 * j + (lane >> 3); srcB[j] = B + (n0 + row) * g.ldb + kbeg + 2 * ((lane & 7) ^ ((row >> 1) & 7)); dstB[j] = ASZ + (16 * wave + 8 * j) * DBK; }
    }
    stepA = AKM ? (int64_t)DBK * g.lda : DBK;
    stepB = BKM ? (int64_t)DBK * g.ldb : DBK;
#define D_ISSUE(kb, SLOT)                                                                                                           \
    do {                                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[j] + (kb) * stepA),               \
                                             (__attribute__((address_space(3))) void*)(&smem[SLOT][dstA[j]]), 16, 0, 0);            \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[j] + (kb) * stepB),               \
                                             (__attribute__((address_space(3))) void*)(&smem[SLOT][dstB[j]]), 16, 0, 0);            \
        }                                                                                                                           \
    } while (0)
#define D_WAIT(N) do { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    const int li = lane & 15, lk = lane >> 4;
    // operand element (row, k = 4 q + lk) of a slab
    int offA[4][4], offB[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int row = wm + 16 * x + li;
            offA[x][q] = AKM ? (4 * q + lk) * DKM_LD + row : row * DBK + ((((4 * q + lk) >> 1) ^ ((row >> 1) & 7)) << 1) + (lk & 1);
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int row = wn + 16 * y + li;
            offB[y][q] = ASZ + (BKM ? (4 * q + lk) * DKM_LD + row : row * DBK + ((((4 * q + lk) >> 1) ^ ((row >> 1) & 7)) << 1) + (lk & 1));
        }
    }
#define D_COMPUTE(SLOT)                                                                                                             \
    do {                                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                             \
            double a_[4], b_[2];                                                                                                    \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) a_[x] = smem[SLOT][offA[x][q]];                                           \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) b_[y] = smem[SLOT][offB[y][q]];                                           \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                                           \
                _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                       \
                    acc.c[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[x], b_[y], acc.c[x][y], 0, 0, 0);                         \
        }                                                                                                                           \
    } while (0)
    // block kk in SLOT; block kk + 2 (clamped to the last block: its surplus copies are never read) is requested into the slot block
    // kk - 1 left at the last barrier; then the MFMAs; then the wait for block kk + 1 with block kk + 2 in flight (4 requests per block)
#define D_STEP(kk, SLOT, SLOT2)                                                                                                     \
    do {                                                                                                                            \
        const int64_t k2_ = (kk) + 2 < nk ? (kk) + 2 : nk - 1;                                                                      \
        D_ISSUE(k2_, SLOT2);                                                                                                        \
        D_COMPUTE(SLOT);                                                                                                            \
        D_WAIT(4);                                                                                                                  \
    } while (0)
    if (nk > 0) {
        int64_t kb = 0;
        for (int i = (int)(nk % 3); i > 0; --i, ++kb) {        // the remainder first, unpipelined
            D_ISSUE(kb, 0);
            D_WAIT(0);
            D_COMPUTE(0);
            __builtin_amdgcn_s_barrier();
        }
        if (kb < nk) {
            D_ISSUE(kb, 0);
            D_ISSUE(kb + 1, 1);
            D_WAIT(4);
            for (; kb < nk; kb += 3) {
                D_STEP(kb, 0, 2);
                D_STEP(kb + 1, 1, 0);
                D_STEP(kb + 2, 2, 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#undef D_STEP
#undef D_COMPUTE
#undef D_WAIT
#undef D_ISSUE
    const double alpha = g.alpha, beta = g.beta;
    const bool atomic = g.atomic != 0;
    acc.for_each(wm, wn, lane, [&](int r, int c, double v) {
        const int64_t row = m0 + r, col = n0 + c;
        if (!(g.lower_only && col > row)) {
            double* p = C + row * g.ldc + col;
            if (atomic) atomic_add(p, alpha * v);
            else *p = (beta == 0.0) ? alpha * v : alpha * v + beta * (*p);
        }
    });
}

// C *= beta (or C = 0) ahead of a split-K launch whose epilogue is atomicAdd
template <typename T>
__global__ void scale_kernel(T* C, int64_t M, int64_t N, int64_t ldc, int64_t sC, T beta, int lower_only) {
    T* c = C + (int64_t)blockIdx.z * sC;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col < N && !(lower_only && col > row)) {
        T* p = c + row * ldc + col;
        *p = (beta == (T)0) ? (T)0 : beta * (*p);
    }
}

// ---- small-tile float64 variant for the latency-bound (M x M) core --------------------------------------------------------------------
// 64 x 64 block tile, BK = 16, 4 waves of 32 x 32 (2 x 2 v_mfma_f64_16x16x4_f64), 34 KB of LDS and < 64 VGPRs: four of these fit a CU NEXT
// TO the big streaming kernels (the 128 x 128 kernel needs a whole CU's LDS and registers, so its workgroups used to wait for a CU to
// drain), 4x more tiles for a 1024^2 output, and the 64-wide panel updates of potrf waste no half tile.  Guarded scalar loaders only.
constexpr int SBM_ = 64, SBK_ = 16, SLD_ = 66;
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_small_f64_kernel(GemmArgs<double> g) {
    __shared__ double sm[2][2][SBK_ * SLD_];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int64_t wid = blockIdx.x;
    const int64_t zs = wid / g.ntiles;
    const int64_t t = wid % g.ntiles;
    int64_t tile_m, tile_n;
    if (g.lower_only) {
        int64_t row = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        tile_m = row; tile_n = t - row * (row + 1) / 2;
    } else { tile_m = t % g.tm; tile_n = t / g.tm; }
    const int64_t m0 = tile_m * SBM_, n0 = tile_n * SBM_;
    const int batch = (int)(zs / g.splitk), split = (int)(zs % g.splitk);
    const double* __restrict__ A = g.A + (int64_t)batch * g.sA;
    const double* __restrict__ B = g.B + (int64_t)batch * g.sB;
    double* __restrict__ C = g.C + (int64_t)batch * g.sC;
    int64_t kbeg = (int64_t)split * g.kchunk;
    int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    if (g.k_from_m == 1) { const int64_t kf = m0 / SBK_ * SBK_; if (kf > kbeg) kbeg = kf; }
    if (g.k_from_m == 2) { const int64_t kl = m0 + 64; if (kl < kend) kend = kl; }
    f64x4 c[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[x][y][r] = 0.0;
    // loaders: 64 x 16 elements per operand per k tile = 4 per thread.  An operand whose k index is the contiguous one in memory (A not
    // transposed, B transposed) is read as 4 CONSECUTIVE k of one row per thread (mn = tid >> 2, k = 4 (tid & 3) + j): a wave instruction
    // then touches 16 rows x 32 contiguous bytes instead of 64 rows x 8 bytes (the rows are lda apart: every lane its own cache line);
    // the other layout (mn contiguous) keeps element (mn = tid & 63, k = (tid >> 6) + 4 j): 64 consecutive doubles per k.
    const int lmn = tid & 63, lk0 = tid >> 6;          // mn-contiguous operands
    const int qmn = tid >> 2, qk0 = (tid & 3) * 4;     // k-contiguous operands
    double ra[4], rb[4];
    auto load = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (TA) { const int64_t kk = k0 + lk0 + 4 * j, ia = m0 + lmn; ra[j] = (ia < g.M && kk < kend) ? A[kk * g.lda + ia] : 0.0; }
            else { const int64_t kk = k0 + qk0 + j, ia = m0 + qmn; ra[j] = (ia < g.M && kk < kend) ? A[ia * g.lda + kk] : 0.0; }
            if (TB) { const int64_t kk = k0 + qk0 + j, ib = n0 + qmn; rb[j] = (ib < g.N && kk < kend) ? B[ib * g.ldb + kk] : 0.0; }
            else { const int64_t kk = k0 + lk0 + 4 * j, ib = n0 + lmn; rb[j] = (ib < g.N && kk < kend) ? B[kk * g.ldb + ib] : 0.0; }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (TA) sm[buf][0][(lk0 + 4 * j) * SLD_ + lmn] = ra[j]; else sm[buf][0][(qk0 + j) * SLD_ + qmn] = ra[j];
            if (TB) sm[buf][1][(qk0 + j) * SLD_ + qmn] = rb[j]; else sm[buf][1][(lk0 + 4 * j) * SLD_ + lmn] = rb[j];
        }
    };
    if (kbeg < kend) { load(kbeg); store(0); }
    __syncthreads();
    const int li = lane & 15, lkk = lane >> 4;
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += SBK_) {
        const bool more = k0 + SBK_ < kend;
        if (more) load(k0 + SBK_);
        const double* As = sm[cur][0];
        const double* Bs = sm[cur][1];
#pragma unroll
        for (int ks = 0; ks < SBK_; ks += 4) {
            double a[2], b[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) { a[x] = As[(ks + lkk) * SLD_ + wm + 16 * x + li]; b[x] = Bs[(ks + lkk) * SLD_ + wn + 16 * x + li]; }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) c[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], c[x][y], 0, 0, 0);
        }
        if (more) store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const double alpha = g.alpha, beta = g.beta;
    const bool atomic = g.atomic != 0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm + x * 16 + (lane >> 4) + 4 * r, col = n0 + wn + y * 16 + (lane & 15);
                if (row < g.M && col < g.N && !(g.lower_only && col > row)) {
                    double* p = C + row * g.ldc + col;
                    if (atomic) atomic_add(p, alpha * c[x][y][r]);
                    else *p = (beta == 0.0) ? alpha * c[x][y][r] : alpha * c[x][y][r] + beta * (*p);
                }
            }
}

// launch of the small-tile variant; returns false when the problem should take the 128 x 128 kernel instead
template <typename T>
bool gemm_small_launch(mxf_ctx*, GemmArgs<T>&, int, int, int64_t, int64_t, int64_t, double, int, int, hipStream_t, int&) { return false; }
template <>
bool gemm_small_launch<double>(mxf_ctx* h, GemmArgs<double>& g, int ta, int tb, int64_t M, int64_t N, int64_t K, double beta, int batch, int lower_only,
                               hipStream_t st, int& rc) {
    static const int small_env = MXF_KNOB("MXF_GEMM_SMALL", 1);
    const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (!small_env || t128 > 64 || M > 65535) return false;       // the big kernel fills at least a quarter of the chip: keep it
    // r06: a LONG contraction over a small output (the float64 step's Psi2 = Kuf Kuf^T: 1024 x 1024 x 2.1 M) is throughput-bound, not
    // latency-bound -- it belongs on the 128 x 128 LDS-DMA kernel with split-K (probe knob MXF_GEMM_SMALL_KMAX: contraction length above
    // which the small-tile kernel steps aside)
    static const int64_t kmax_env = MXF_KNOB("MXF_GEMM_SMALL_KMAX", 16384);
    if (K > kmax_env && !g.k_from_m && M >= 512 && N >= 512) return false;      // (skinny outputs -- K^T y, M x P -- stay: the 128 x 128 tiles would idle)
    const int64_t tm = (M + SBM_ - 1) / SBM_, tn = (N + SBM_ - 1) / SBM_;
    if (lower_only && tm != tn) return false;
    const int64_t tiles = (lower_only ? tm * (tm + 1) / 2 : tm * tn) * batch;
    int64_t splitk = 1;
    const int64_t slots = 1024;                                      // 4 workgroups per CU
    if (tiles < slots && K >= 128 && !g.k_from_m) { splitk = slots / tiles; const int64_t mx = K / 64 > 0 ? K / 64 : 1; if (splitk > mx) splitk = mx; if (splitk < 1) splitk = 1; }
    int64_t kchunk = SBK_;
    if (K > 0) { kchunk = (K + splitk - 1) / splitk; kchunk = (kchunk + SBK_ - 1) / SBK_ * SBK_; splitk = (K + kchunk - 1) / kchunk; } else splitk = 1;
    g.splitk = (int)splitk; g.kchunk = kchunk; g.atomic = splitk > 1;
    g.tm = tm; g.tn = tn; g.ntiles = lower_only ? tm * (tm + 1) / 2 : tm * tn; g.nwg = g.ntiles * batch * splitk;
    g.xc_max = 0; g.rev_m = 0;
    rc = 0;
    if (g.atomic && beta != 1.0) {
        dim3 gs((unsigned)((N + 255) / 256), (unsigned)M, (unsigned)batch);
        hipLaunchKernelGGL((scale_kernel<double>), gs, dim3(256), 0, st, g.C, M, N, g.ldc, g.sC, beta, lower_only);
    }
    dim3 grid((unsigned)g.nwg, 1, 1);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_small_f64_kernel<false, false>), grid, dim3(256), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_small_f64_kernel<false, true>), grid, dim3(256), 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_small_f64_kernel<true, false>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_small_f64_kernel<true, true>), grid, dim3(256), 0, st, g);
    if (hipGetLastError() != hipSuccess) rc = -5;
    return true;
}

template <typename T>
int gemm_typed(mxf_ctx* h, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda,
               int64_t sA, const void* B, int64_t ldb, int64_t sB, double beta, void* C, int64_t ldc, int64_t sC,
               int batch, int lower_only, hipStream_t st, int reserve_cus, int k_from_m) {
    GemmArgs<T> g;
    g.k_from_m = (k_from_m == 1 && ta) ? 1 : ((k_from_m == 2 && !ta) ? 2 : 0);     // the caller vouches for the triangular operand
    g.A = (const T*)A; g.B = (const T*)B; g.C = (T*)C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
    g.alpha = (T)alpha; g.beta = (T)beta; g.lower_only = lower_only;
    {
        constexpr int VEC = Vec16<T>::n;
        auto ok = [&](const void* p, int64_t ld, int64_t st) { return ((uintptr_t)p % 16 == 0) && (ld % VEC == 0) && (st % VEC == 0); };
        g.vecA = ok(A, lda, sA); g.vecB = ok(B, ldb, sB);
    }
    {   // latency-bound float64 products of the (M x M) core: small-tile variant
        int rc_small = 0;
        if (gemm_small_launch<T>(h, g, ta, tb, M, N, K, beta, batch, lower_only, st, rc_small)) {
            if (rc_small) MXF_FAIL(h, -5, "mxf_gemm: small-tile launch failed");
            return 0;
        }
    }
    const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    // split K when the output grid cannot fill 256 CUs and K is long
    const int64_t tiles = (lower_only ? tm * (tm + 1) / 2 : tm * tn) * batch;   // EXACT count (an estimate of 32 for 36 tiles cost 35 %)
    // split K when the output grid cannot fill the chip.  Resident slots: 256 CUs x (2 workgroups f32 | 1 f64); pick the
    // split so that the grid is (just under) a whole number of rounds -- 576 workgroups on 512 slots would run a
    // half-empty second round (measured: Psi2 33 ms -> 19 ms with 504).
    int splitk = 1;
    // reserve_cus: leave that many CUs without a workgroup of this (register-saturating) kernel, so that latency-bound
    // kernels of a concurrent stream can still be scheduled (the split-K grid is sized to the remaining CUs)
    const int64_t slots = (256 - reserve_cus) * (sizeof(T) == 4 ? 2 : 1);
    if (tiles < slots && K >= 256 && !g.k_from_m) {
        const int64_t maxsplit = K / 128 > 0 ? K / 128 : 1;     // small latency-bound GEMMs of the (M x M) core split too
        int64_t sk = slots / tiles;
        if (sk * tiles < (slots * 3) / 4) sk = (2 * slots) / tiles;   // one round would leave >25% of the slots idle: use two
        if (sk > maxsplit) sk = maxsplit;
        if (sk < 1) sk = 1;
        splitk = (int)sk;
    }
    {   // probe knob (A/B experiments only)
        static const int sk_env = MXF_KNOB("MXF_GEMM_SPLITK", 0);
        if (sk_env > 0 && K >= 4096) splitk = sk_env;
    }
    int64_t kchunk = BK;
    if (K > 0) {
        kchunk = (K + splitk - 1) / splitk;
        kchunk = (kchunk + BK - 1) / BK * BK;
        splitk = (int)((K + kchunk - 1) / kchunk);
    } else {
        splitk = 1;      // empty contraction: C = beta C
    }
    g.splitk = splitk; g.kchunk = kchunk; g.atomic = splitk > 1;
    if (lower_only && tm != tn) MXF_FAIL(h, -2, "mxf_gemm: lower_only needs a square output");
    g.tm = tm; g.tn = tn;
    g.ntiles = lower_only ? tm * (tm + 1) / 2 : tm * tn;
    g.nwg = g.ntiles * batch * splitk;
    g.xc_max = 0;
    static const int rev_env = MXF_KNOB("MXF_GEMM_REV_M", 1);
    g.rev_m = rev_env;
    static const int tri_balance = MXF_KNOB("MXF_GEMM_TRI_BALANCE", 1);
    if (tri_balance && lower_only && g.k_from_m == 1 && splitk == 1 && tm >= 16) {
        for (int64_t x = 0; x < 8; ++x) {
            int64_t cnt = 0;
            for (int64_t r = x; r < tm; r += 8) cnt += r + 1;
            if (cnt > g.xc_max) g.xc_max = cnt;
        }
        g.nwg = 8 * g.xc_max * batch;
    }
    if (g.nwg > 2147483647LL) MXF_FAIL(h, -3, "mxf_gemm: grid too large");
    if (g.atomic && beta != 1.0) {     // beta == 1 (the potrf / trsm updates): C is accumulated into as is, nothing to pre-scale
        dim3 gs((unsigned)((N + 255) / 256), (unsigned)M, (unsigned)batch);
        if (M > 65535) MXF_FAIL(h, -3, "mxf_gemm: split-K path needs M<=65535");
        hipLaunchKernelGGL((scale_kernel<T>), gs, dim3(256), 0, st, (T*)C, M, N, ldc, sC, (T)beta, lower_only);
    }
    dim3 grid((unsigned)g.nwg, 1, 1);
    if constexpr (sizeof(T) == 8) {
        static const int dma_env = MXF_KNOB("MXF_GEMM_F64_DMA", 1);
        if (dma_env && g.vecA && g.vecB && M % BM == 0 && N % BN == 0 && K % DBK == 0 && kchunk % DBK == 0 && NWAVE == 8) {
            if (!ta && tb) hipLaunchKernelGGL((gemm_f64_dma_kernel<false, false>), grid, dim3(512), 0, st, g);
            else if (ta && !tb) hipLaunchKernelGGL((gemm_f64_dma_kernel<true, true>), grid, dim3(512), 0, st, g);
            else if (ta && tb) hipLaunchKernelGGL((gemm_f64_dma_kernel<true, false>), grid, dim3(512), 0, st, g);
            else hipLaunchKernelGGL((gemm_f64_dma_kernel<false, true>), grid, dim3(512), 0, st, g);
            MXF_LAUNCH_CHECK(h);
            return 0;
        }
    }
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(NT), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(NT), 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(NT), 0, st, g);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

}  // namespace

int mxf_gemm_internal(mxf_ctx* h, int dtype, int ta, int tb, int64_t M, int64_t N, int64_t K, double alpha,
                      const void* A, int64_t lda, int64_t sA, const void* B, int64_t ldb, int64_t sB, double beta,
                      void* C, int64_t ldc, int64_t sC, int batch, int lower_only, hipStream_t st, int reserve_cus, int k_from_m) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (dtype == MXF_F32) return gemm_typed<float>(h, ta, tb, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch, lower_only, st, reserve_cus, k_from_m);
    if (dtype == MXF_F64) return gemm_typed<double>(h, ta, tb, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch, lower_only, st, reserve_cus, k_from_m);
    MXF_FAIL(h, -2, "mxf_gemm: bad dtype %d", dtype);
}

extern "C" int mxf_gemm(mxf_handle h, int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                        double alpha, const void* A, int64_t lda, int64_t strideA,
                        const void* B, int64_t ldb, int64_t strideB,
                        double beta, void* C, int64_t ldc, int64_t strideC, int batch, void* stream) {
    if (!h) return -1;
    if (M < 0 || N < 0 || K < 0 || batch < 0) MXF_FAIL(h, -2, "mxf_gemm: negative dimension");
    if ((M > 0 && N > 0 && batch > 0) && (!A || !B || !C) && K > 0) MXF_FAIL(h, -2, "mxf_gemm: null operand");
    static const int lower_env = MXF_KNOB("MXF_GEMM_LOWER", 0);   // probe knob
    return mxf_gemm_internal(h, dtype, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc,
                             strideC, batch, (lower_env && M == N) ? 1 : 0, (hipStream_t)stream);
}

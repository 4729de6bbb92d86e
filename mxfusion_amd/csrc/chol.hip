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

// 1/sqrt(d) to full precision: hardware v_rsq estimate + Newton steps (cheaper than a correctly-rounded sqrt AND a division)
__device__ __forceinline__ float inv_sqrt(float d) { float r = rsqrtf(d); r = r * (1.5f - 0.5f * d * r * r); return r; }
__device__ __forceinline__ double inv_sqrt(double d) {
    // (the pivots of a positive-definite matrix that is worth factoring are normal numbers: no scaling of the argument; the hardware
    //  estimate is good to ~2^-26, one third-order correction y (1 + e/2 + 3 e^2/8), e = 1 - d y^2, leaves < 1 ulp -- five dependent
    //  operations on the critical path of every column instead of the library routine's special-case handling plus a Newton step)
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
constexpr int NBO = 512;   // outer panel
constexpr int TRI_MIN = 1024;   // trtri merge levels from this block size on use the triangular-aware GEMM k ranges

__device__ __forceinline__ float readlane_t(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double readlane_t(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ---- fused panel step: every workgroup re-factors the 64x64 diagonal block in LDS (rank-4 blocked: 16 barrier pairs instead of
// 64) and then solves its own 128 rows against it; workgroup 0 owns the diagonal block itself.  One launch per 64-wide block column
// instead of two (diag + solve), and no dependent launch gap between them.
template <typename T>
__global__ __launch_bounds__(128) void potrf_panel_kernel(T* __restrict__ A, int64_t lda, int64_t sA, int64_t k0, int nb, int64_t n,
                                                           int* __restrict__ info, int* __restrict__ arrived) {
    __shared__ T a[NB][NB + 1];
    __shared__ T invd[NB];
    __shared__ T t[128][NB + 1];
    const int tid = threadIdx.x, b = blockIdx.y;
    T* Ab = A + (int64_t)b * sA;
    T* D = Ab + k0 * lda + k0;
    {   // all 32 loads of a thread are issued before the first LDS store: one memory round trip instead of one per unrolled batch
        // (timestamps: 11.5 us for this 32 KB block and 19 us for the 64 KB row tile below, of a 64 us panel step, before)
        T va[NB * NB / 128];
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) {
            const int e = tid + it * 128, i = e / NB, c = e % NB;
            T v = (T)0;
            if (i < nb && c < nb) { if (c <= i) v = D[(int64_t)i * lda + c]; }
            else if (i == c) v = (T)1;          // identity padding of a ragged last block
            va[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NB * NB / 128; ++it) { const int e = tid + it * 128; a[e / NB][e % NB] = va[it]; }
    }
    __syncthreads();
    // Every workgroup re-factors the diagonal block from the UNFACTORED values in global memory, and workgroup 0 writes the factor back
    // in place: it may do so only after all the others have taken their copy (they can start arbitrarily late when other kernels hold
    // the CUs).  Arrival counter: one per batch item, zero on entry, reset to zero by workgroup 0.
    if (blockIdx.x != 0 && tid == 0) { __threadfence(); atomicAdd(arrived + b, 1); }
    // 64 x 64 diagonal block, 16 columns at a time: (1) the 16 x 16 sub-block by ONE wave with a row per lane in registers (pivots and
    // columns travel by v_readlane: no LDS round trip, no barrier), (2) the rows below it by substitution (one row per thread, x[16]),
    // (3) the trailing lower triangle by MFMA:  A22 -= X X^T.  (The rank-4 LDS version it replaces spent ~40 us in 32 barriers and
    // dependent LDS round trips.)
    {
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
        for (int blk = 0; blk < NB; blk += 16) {
            if (wave == 0) {                                   // (1)
                T r[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) r[c] = a[blk + li][blk + c];       // lanes 16..63 mirror lanes 0..15 (harmless)
                int bad = -1;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    T d = readlane_t(r[j], j);
                    if (!(d > (T)0)) { if (bad < 0) bad = j; d = (T)1; }        // not positive definite (or NaN): record, keep going finite
                    const T inv = inv_sqrt(d);
                    const T l = (li == j) ? d * inv : r[j] * inv;
                    r[j] = l;
                    if (lane == 0) invd[blk + j] = inv;
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) r[c] = fma(-l, readlane_t(l, c), r[c]);
                }
                if (bad >= 0 && lane == 0 && blockIdx.x == 0 && blk + bad < nb && info && info[b] == 0) info[b] = (int)(k0 + blk + bad + 1);
                if (lane < 16) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) a[blk + lane][blk + c] = (c <= lane) ? r[c] : (T)0;
                }
            }
            __syncthreads();
            if (blk + 16 < NB) {
                const int i = blk + 16 + tid;                   // (2) rows below the sub-block: X L_bb^T = A_ib
                if (i < NB) {
                    T x[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        T s0 = a[i][blk + c], s1 = (T)0;
#pragma unroll
                        for (int k = 0; k + 1 < c; k += 2) { s0 = fma(-x[k], a[blk + c][blk + k], s0); s1 = fma(-x[k + 1], a[blk + c][blk + k + 1], s1); }
                        if (c & 1) s0 = fma(-x[c - 1], a[blk + c][blk + c - 1], s0);
                        x[c] = (s0 + s1) * invd[blk + c];
                    }
#pragma unroll
                    for (int c = 0; c < 16; ++c) a[i][blk + c] = x[c];
                }
                __syncthreads();
                // (3) trailing update, lower tiles (I0 >= J0) of 16 x 16, dealt to the two waves
                int tix = 0;
                for (int I0 = blk + 16; I0 < NB; I0 += 16)
                    for (int J0 = blk + 16; J0 <= I0; J0 += 16, ++tix) {
                        if ((tix & 1) != wave) continue;
                        T af[4], bf[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { af[q] = -a[I0 + li][blk + 4 * q + lq]; bf[q] = a[J0 + li][blk + 4 * q + lq]; }
                        if constexpr (sizeof(T) == 8) {
                            typedef double f64x4_ __attribute__((ext_vector_type(4)));
                            f64x4_ cf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cf[r] = a[I0 + lq + 4 * r][J0 + li];
#pragma unroll
                            for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], bf[q], cf, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) a[I0 + lq + 4 * r][J0 + li] = cf[r];
                        } else {
                            typedef float f32x4_ __attribute__((ext_vector_type(4)));
                            f32x4_ cf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cf[r] = a[I0 + lq * 4 + r][J0 + li];
#pragma unroll
                            for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], bf[q], cf, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) a[I0 + lq * 4 + r][J0 + li] = cf[r];
                        }
                    }
                __syncthreads();
            }
        }
    }
    if (blockIdx.x == 0) {
        if (tid == 0) {
            const int others = (int)gridDim.x - 1;
            while (__hip_atomic_load(arrived + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < others) __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store(arrived + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int e = tid; e < nb * nb; e += 128) {
            const int i = e / nb, c = e % nb;
            D[(int64_t)i * lda + c] = (c <= i) ? a[i][c] : (T)0;     // MXNet potrf zeroes the strict upper part
        }
        return;
    }
    // rows below: X L11^T = A21, one row per lane
    const int64_t r0 = k0 + nb, nrows = n - r0;
    const int64_t rb = r0 + (int64_t)(blockIdx.x - 1) * 128;
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // two batches of 32 loads per thread, each issued back to back
        T vt[NB / 2];
#pragma unroll
        for (int it = 0; it < NB / 2; ++it) {
            const int e = tid + (half * (NB / 2) + it) * 128, r = e / NB, c = e % NB;
            vt[it] = (rb + r < r0 + nrows && c < nb) ? Ab[(rb + r) * lda + k0 + c] : (T)0;
        }
#pragma unroll
        for (int it = 0; it < NB / 2; ++it) { const int e = tid + (half * (NB / 2) + it) * 128; t[e / NB][e % NB] = vt[it]; }
    }
    __syncthreads();
    // Blocked substitution: 16 columns at a time by plain substitution (one row per lane, x[16] in registers), then the remaining columns of
    // the tile are updated with MFMA:  T[:, J] -= X[:, blk] L11[J, blk]^T.  The un-blocked form (one row per lane, 64 dependent steps,
    // 2016 broadcast LDS reads with x[64] pinning 128 VGPRs) kept only two LDS reads in flight and took ~40 us of the 84 us kernel.
    {
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
        for (int blk = 0; blk < NB; blk += 16) {
            T x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                T s0 = t[tid][blk + c], s1 = (T)0;          // two accumulators: halves the dependent-FMA chain
#pragma unroll
                for (int k = 0; k + 1 < c; k += 2) { s0 = fma(-x[k], a[blk + c][blk + k], s0); s1 = fma(-x[k + 1], a[blk + c][blk + k + 1], s1); }
                if (c & 1) s0 = fma(-x[c - 1], a[blk + c][blk + c - 1], s0);
                x[c] = (s0 + s1) * invd[blk + c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) t[tid][blk + c] = x[c];
            __syncthreads();
            if (blk + 16 < NB) {
                for (int rt = wave * 4; rt < wave * 4 + 4; ++rt) {          // 8 row tiles of 16 rows, 4 per wave
                    const int rr = rt * 16;
                    T af[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) af[q] = -t[rr + li][blk + 4 * q + lq];
                    for (int J0 = blk + 16; J0 < NB; J0 += 16) {
                        if constexpr (sizeof(T) == 8) {
                            typedef double f64x4_ __attribute__((ext_vector_type(4)));
                            f64x4_ cf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cf[r] = t[rr + lq + 4 * r][J0 + li];
#pragma unroll
                            for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], a[J0 + li][blk + 4 * q + lq], cf, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) t[rr + lq + 4 * r][J0 + li] = cf[r];
                        } else {
                            typedef float f32x4_ __attribute__((ext_vector_type(4)));
                            f32x4_ cf;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cf[r] = t[rr + lq * 4 + r][J0 + li];
#pragma unroll
                            for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q], a[J0 + li][blk + 4 * q + lq], cf, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) t[rr + lq * 4 + r][J0 + li] = cf[r];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll 8
    for (int e = tid; e < 128 * NB; e += 128) {
        const int r = e / NB, c = e % NB;
        if (rb + r < r0 + nrows && c < nb) Ab[(rb + r) * lda + k0 + c] = t[r][c];
    }
}

// ---- tile-dataflow Cholesky (float64, n a multiple of 64): one workgroup per 64-row block row -----------------------------------------
// n <= 512 in ONE launch; larger n one launch per 512-column outer panel (below).  The launch-per-panel form above costs 16 x (panel
// kernel ~45 us + left-looking GEMM + two dependent-launch gaps) at n = 1024: 0.86 ms, every one of them on the critical path of the SVGP
// step when few samples are left per GPU.  Here workgroup i owns block row i and walks its tiles j = 0 .. i
// left-looking:
//     C = A[i][j] - sum_{k<j} L[i][k] L[j][k]^T      (ONE k loop of length 64 j over two contiguous row panels, f64 MFMA, software-pipelined)
//     j < i :  L[i][j] = C L[j][j]^-T                 (with the inverses of L[j][j]'s four 16 x 16 diagonal blocks, which its owner publishes)
//     j = i :  L[i][i] = chol(C)                      (16 x 16 sub-blocks as rank-1 MFMA steps in one accumulator)
// and hands tiles on through a progress counter per block row (release store after the tile is in memory; readers acquire): tile (i, j)
// needs progress[j] >= j before its k loop and progress[j] == j + 1 before its solve.  The diagonal tile's update is accumulated
// incrementally (one 64^3 product after every solve), so the critical path per block column is factor -> solve of the next row's tile ->
// one product -> factor: 23 us (DESIGN.md section 4 has the stage table; tests/probes/potrf_trace.py measures it).  Every workgroup
// must be resident at once (<= 256 workgroups); spins are bounded (a lost hand-off reports info = -1 instead of hanging the queue).
#ifdef MXF_POTRF_TRACE
// probe build only (tests/probes/potrf_trace.py): 100 MHz timestamps of the stages of block rows 0 .. 15, [row][column j][stage]
__device__ long long potrf_trace_buf[16 * 17 * 16];
#define PT_STAMP(k) do { if (tid == 0 && i < 16) potrf_trace_buf[(i * 17 + jt) * 16 + (k)] = wall_clock64(); } while (0)
extern "C" int mxf_debug_potrf_trace(long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(potrf_trace_buf), sizeof(long long) * 16 * 17 * 16); }
#else
#define PT_STAMP(k) do { } while (0)
#endif
constexpr int PT_SLD = 66;      // LDS row stride of the staged k chunks (as the small GEMM)
typedef double pt_f64x4 __attribute__((ext_vector_type(4)));

// Panel mode (large n): the same kernel factors ONE outer panel -- columns [c0, c0 + 64 npt), rows c0 .. n, one workgroup per block row
// below c0 -- left-looking INSIDE the panel (columns left of c0 were applied by the trailing updates of the earlier panels): block rows
// i < npt end with their diagonal tile, the others only solve.  nbr = block rows, counters: nbr progress words + one completion word.
__global__ __launch_bounds__(256) void potrf_tiles_kernel(double* __restrict__ A, int64_t lda, int64_t sA, int64_t c0, int npt, int* __restrict__ info,
                                                          int* __restrict__ progress_all, double* __restrict__ inv_all, int row0, int nowait) {
    __shared__ double a[NB][NB + 1];          // diagonal factor L[j][j] (or, for j == i, the tile being factored)
    __shared__ double invd[NB];
    __shared__ double minv[4][16][17];       // the arrived diagonal factor's four inverse blocks
    // the tile being solved, and (never at the same time) the k chunks of the two row panels of the left-looking product: 70 KB of LDS in
    // all, so that a 64 KB float64 GEMM workgroup of the look-ahead trailing update still fits on the same CU
    __shared__ double tsm[(2 * 2 * 16 * PT_SLD > NB * (NB + 1)) ? 2 * 2 * 16 * PT_SLD : NB * (NB + 1)];
    double (*t)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(tsm);
    double (*sm)[2][16 * PT_SLD] = reinterpret_cast<double (*)[2][16 * PT_SLD]>(tsm);
    __shared__ int sflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    // row0 / nowait: the SECOND launch of a split outer panel -- block rows row0 .. below an already factored diagonal block: every tile
    // they need is final (stream order), nothing is waited for or published
    const int i = row0 + (int)blockIdx.x, b = blockIdx.y, nbk = (int)gridDim.x;
    double* Ab = A + (int64_t)b * sA;
    int* progress = progress_all + (int64_t)b * (nbk + 1);
    double* invs = inv_all + (int64_t)b * npt * 1024;     // [block row][16-column block][row][column] inverses of the diagonal factors' 16 x 16 diagonal blocks
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    bool lost = false;

    // all threads: block until progress[row] >= need; returns the value seen (progress only grows: the caller skips later waits it covers)
    auto wait_for = [&](int row, int need) -> int {
        if (nowait) return 1 << 30;
        if (tid == 0) {
            int spins = 0, v;
            while ((v = __hip_atomic_load(progress + row, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) { v = -1; break; }
            }
            sflag = v;
        }
        __syncthreads();
        const int v = sflag;
        if (v < 0) lost = true;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);         // every wave: drop stale lines before reading the published tiles
        __syncthreads();                                  // (sflag is rewritten by the next wait)
        return v < 0 ? (1 << 30) : v;
    };
    // C (this wave's 32 x 32 quadrant, accumulator layout) += P[rows r0 ..][k0 .. k1) . Q[rows q0 ..][k0 .. k1)^T, both row panels of A;
    // k0, k1 multiples of 64.  One wave per SIMD: nothing hides a latency unless the code does.  Chunks of 16 columns; FOUR chunks of
    // global loads in flight (the tiles come from other CUs' stores, ~2 us away); two LDS buffers and two operand register sets: while
    // the 16 MFMAs of chunk c issue, chunk c + 1 goes registers -> LDS -> barrier -> operand registers in between them.
    auto panel_product = [&](pt_f64x4 (&c)[2][2], int64_t r0, int64_t q0, int64_t k0, int64_t k1, int gate_row) {
        const int qmn = tid >> 2, qk0 = (tid & 3) * 4;
        double ra[4][4], rb[4][4], oa[2][4][2], ob[2][4][2];
        if (k0 >= k1) return;
        int seen = 0;
        auto gate = [&](int64_t k) {                       // the 64-column tile of row gate_row that starts at column k
            if (gate_row < 0 || k >= k1) return;
            const int need = (int)((k - c0) / NB) + 1;
            if (seen < need) seen = wait_for(gate_row, need);
        };
        const double* pa = Ab + (r0 + qmn) * lda + qk0;
        const double* pb = Ab + (q0 + qmn) * lda + qk0;
        const int64_t C = (k1 - k0) / 16;
#define PT_LOAD(s, k) do { _Pragma("unroll") for (int j = 0; j < 4; ++j) { ra[s][j] = pa[(k) + j]; rb[s][j] = pb[(k) + j]; } } while (0)
#define PT_STORE(s, b) do { _Pragma("unroll") for (int j = 0; j < 4; ++j) { sm[b][0][(qk0 + j) * PT_SLD + qmn] = ra[s][j]; sm[b][1][(qk0 + j) * PT_SLD + qmn] = rb[s][j]; } } while (0)
#define PT_READ(b, o) do { _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int x = 0; x < 2; ++x) {                     \
            oa[o][q][x] = sm[b][0][(4 * q + lq) * PT_SLD + wm + 16 * x + li]; ob[o][q][x] = sm[b][1][(4 * q + lq) * PT_SLD + wn + 16 * x + li]; } } while (0)
#define PT_MMA(o, q0_, q1_) do { _Pragma("unroll") for (int q = (q0_); q < (q1_); ++q) _Pragma("unroll") for (int x = 0; x < 2; ++x)          \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) c[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[o][q][x], ob[o][q][y], c[x][y], 0, 0, 0); } while (0)
        gate(k0);
        PT_LOAD(0, k0); PT_LOAD(1, k0 + 16); PT_LOAD(2, k0 + 32); PT_LOAD(3, k0 + 48);
        gate(k0 + 64);
        PT_STORE(0, 0);
        __syncthreads();
        if (C > 4) PT_LOAD(0, k0 + 64);
        PT_READ(0, 0);
        for (int64_t m = 0; m < C; m += 4) {
            gate(k0 + 16 * (m + 8));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int64_t cc = m + s;
                const bool nxt = cc + 1 < C;
                if (nxt) PT_STORE((s + 1) & 3, (s + 1) & 1);
                PT_MMA(s & 1, 0, 2);
                if (nxt) {
                    __syncthreads();
                    if (cc + 5 < C) PT_LOAD((s + 1) & 3, k0 + 16 * (cc + 5));
                    PT_READ((s + 1) & 1, (s + 1) & 1);
                }
                PT_MMA(s & 1, 2, 4);
            }
        }
#undef PT_LOAD
#undef PT_STORE
#undef PT_READ
#undef PT_MMA
    };
    auto zero = [&](pt_f64x4 (&c)[2][2]) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) c[x][y] = pt_f64x4{0.0, 0.0, 0.0, 0.0};
    };
    // accumulator element (x, y, r) <-> tile element (row wm + 16 x + lq + 4 r, column wn + 16 y + li)
    const int64_t ri = c0 + (int64_t)i * NB;
    pt_f64x4 cd[2][2];          // running sum_k L[i][k] L[i][k]^T of the diagonal tile
    zero(cd);
    const int jend = i < npt ? i : npt;
    pt_f64x4 dg[2][2];          // A[i][i] itself, fetched now: the factorisation of the diagonal tile is the critical path of its block column
    if (i < npt) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) dg[x][y][r] = Ab[(ri + wm + 16 * x + lq + 4 * r) * lda + ri + wn + 16 * y + li];
    }

    for (int j = 0; j < jend; ++j) {
        const int64_t rj = c0 + (int64_t)j * NB;
        pt_f64x4 c[2][2];
        zero(c);
        [[maybe_unused]] const int jt = j;
        PT_STAMP(0);
        panel_product(c, ri, rj, c0, rj, j);                      // sum_{k<j} L[i][k] L[j][k]^T  (row j's tiles k < j: gated on progress[j])
        PT_STAMP(1);
        __syncthreads();                                           // (the product's last chunk has been read: its LDS becomes the tile)
        // (the tile's own entries do not depend on L[j][j]: they are in LDS before it arrives)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = wm + 16 * x + lq + 4 * r, cc = wn + 16 * y + li;
                    t[rr][cc] = Ab[(ri + rr) * lda + rj + cc] - c[x][y][r];
                }
        wait_for(j, j + 1);                                        // L[j][j] and the inverses of its four 16 x 16 diagonal blocks
        PT_STAMP(2);
        pt_f64x4 mreg[4];                                          // this lane's entries of the inverses, as MFMA A operands: M_b[li][4 r + lq]
        {
            double lv[16];                                         // 16 independent loads in flight, then LDS
            const int r0 = tid >> 6, cc = tid & 63;
#pragma unroll
            for (int u = 0; u < 16; ++u) lv[u] = Ab[(rj + r0 + 4 * u) * lda + rj + cc];
            double mvv[4];                                         // the four inverses: 1024 contiguous values, through LDS as well
#pragma unroll
            for (int u = 0; u < 4; ++u) mvv[u] = invs[(int64_t)j * 1024 + tid + 256 * u];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[r0 + 4 * u][cc] = (cc <= r0 + 4 * u) ? lv[u] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) minv[u][tid >> 4][tid & 15] = mvv[u];
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mreg[bb][r] = minv[bb][li][4 * r + lq];
        PT_STAMP(3);
        // X L[j][j]^T = T, one 16-row tile per wave, entirely in registers (accumulator layout of the TRANSPOSED tile: lane (li, lq),
        // register r = element (row li, column lq + 4 r) of a 16 x 16 block): X_b = Y_b M_b^T with the published inverse M_b of the
        // diagonal block (4 MFMAs) instead of 16 dependent substitution steps, then Y_b'' -= X_b L_b''b^T for the blocks to the right.
        {
            pt_f64x4 tacc[4];
            const int rw = wave * 16;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) tacc[bb][r] = t[rw + li][16 * bb + lq + 4 * r];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                pt_f64x4 xs = pt_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) xs = __builtin_amdgcn_mfma_f64_16x16x4f64(mreg[bb][r], tacc[bb][r], xs, 0, 0, 0);
#pragma unroll
                for (int b2 = bb + 1; b2 < 4; ++b2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tacc[b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[16 * b2 + li][16 * bb + 4 * r + lq], xs[r], tacc[b2], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) t[rw + li][16 * bb + lq + 4 * r] = xs[r];
            }
        }
        __syncthreads();
        PT_STAMP(4);
        {
            const int r0 = tid >> 6, cc = tid & 63;
#pragma unroll
            for (int u = 0; u < 16; ++u) Ab[(ri + r0 + 4 * u) * lda + rj + cc] = t[r0 + 4 * u][cc];
        }
        // the diagonal tile's share of this column, straight from the solved tile in LDS: cd += L[i][j] L[i][j]^T (the stores above drain
        // meanwhile; the tile is published after it -- the rows below need it for their NEXT column's product only)
        if (i < npt) {
#pragma unroll 4
            for (int ks = 0; ks < NB; ks += 4) {
                double av[2], bv[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) { av[x] = t[wm + 16 * x + li][ks + lq]; bv[x] = t[wn + 16 * x + li][ks + lq]; }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) cd[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], cd[x][y], 0, 0, 0);
            }
        }
        PT_STAMP(5);
        __threadfence();
        __syncthreads();                                           // (also: t is rewritten by the next tile)
        if (tid == 0 && !nowait) __hip_atomic_store(progress + i, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        PT_STAMP(6);
    }
    // M_b = L_bb^-1 of the 16 x 16 block at a[blk ..][blk ..] (one wave; straight to the hand-off scratch): row by row, M[c][:] = W[c][:] / l_cc,
    // then W[R][:] -= L[R][c] M[c][:] for the rows below as ONE rank-1 MFMA; accumulator lane (li, lq), register r = element (lq + 4 r, li).
    auto inverse16 = [&](int blk) {
        pt_f64x4 lb, wv, mv = pt_f64x4{0.0, 0.0, 0.0, 0.0};
        double iv[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) { lb[r] = a[blk + li][blk + lq + 4 * r]; wv[r] = (lq + 4 * r == li) ? 1.0 : 0.0; }
#pragma unroll
        for (int c = 0; c < 16; ++c) iv[c] = invd[blk + c];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int p = c >> 2, sg = c & 3;
            const double msel = (lq == sg) ? 1.0 : 0.0;
            const double mrow = (wv[p] * iv[c]) * msel;
            mv[p] += mrow;
            if (c < 15) wv = __builtin_amdgcn_mfma_f64_16x16x4f64(-lb[p] * msel, mrow, wv, 0, 0, 0);
        }
        double* dst = invs + (int64_t)i * 1024 + (blk >> 4) * 256 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(lq + 4 * r) * 16] = mv[r];
    };
    if (i < npt) {
    [[maybe_unused]] const int jt = i;
    PT_STAMP(7);
    // diagonal tile: factor A[i][i] - cd
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = wm + 16 * x + lq + 4 * r, cc = wn + 16 * y + li;
                a[rr][cc] = (cc <= rr) ? dg[x][y][r] - cd[x][y][r] : 0.0;
            }
    __syncthreads();
    PT_STAMP(8);
    for (int blk = 0; blk < NB; blk += 16) {
        if (wave == 0) {
            // 16 x 16 sub-block on the matrix pipe: the block sits in ONE MFMA accumulator (lane (li, lq), register r = element
            // (row li, column lq + 4 r), valid for column <= row), and every column step is d -> 1/sqrt(d) -> scale -> ONE rank-1 MFMA
            // (operand: the scaled column in the lanes lq == c % 4, zero elsewhere) instead of 15 readlane + FMA column updates:
            // ~15 instructions on the serial chain of a column instead of ~55.
            pt_f64x4 cv, lv = pt_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[r] = a[blk + li][blk + lq + 4 * r];
            int bad = 99;
            double myinv = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int p = c >> 2, sg = c & 3;
                // (branch-free; a non-positive pivot is reported below and its column left unscaled: the factor stays finite)
                const double d = readlane_t(cv[p], sg * 16 + c);
                bad = (!(d > 0.0) && bad > c) ? c : bad;
                const double rs = inv_sqrt(d);
                const double inv = (d > 0.0) ? rs : 1.0;
                const double m = (lq == sg && li >= c) ? 1.0 : 0.0;       // this column's lanes: rows c .. 15 in lane group c % 4
                const double l = (cv[p] * inv) * m;
                lv[p] += l;                                                // (exactly one non-zero contribution per lane and register)
                myinv = (lane == c) ? inv : myinv;
                if (c < 15) cv = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, cv, 0, 0, 0);
            }
            if (bad < 16 && lane == 0 && info && info[b] == 0) info[b] = (int)(ri + blk + bad + 1);
            if (lane < 16) invd[blk + lane] = myinv;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[blk + li][blk + lq + 4 * r] = (lq + 4 * r <= li) ? lv[r] : 0.0;
        }
        __syncthreads();
        if (blk == 0) PT_STAMP(11);
        if (wave == 1) inverse16(blk);                     // M_b for the solving workgroups, next to wave 0's substitution below
        if (blk + 16 < NB) {
            const int rr = blk + 16 + tid;                 // rows below the sub-block: X L_bb^T = A_ib
            if (rr < NB) {
                double x[16];
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) {
                    double s0 = a[rr][blk + c2], s1 = 0.0;
#pragma unroll
                    for (int k = 0; k + 1 < c2; k += 2) { s0 = fma(-x[k], a[blk + c2][blk + k], s0); s1 = fma(-x[k + 1], a[blk + c2][blk + k + 1], s1); }
                    if (c2 & 1) s0 = fma(-x[c2 - 1], a[blk + c2][blk + c2 - 1], s0);
                    x[c2] = (s0 + s1) * invd[blk + c2];
                }
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) a[rr][blk + c2] = x[c2];
            }
            __syncthreads();
            if (blk == 0) PT_STAMP(12);
            int tix = 0;                                   // trailing lower tiles of 16 x 16, dealt to the four waves
            for (int I0 = blk + 16; I0 < NB; I0 += 16)
                for (int J0 = blk + 16; J0 <= I0; J0 += 16, ++tix) {
                    if ((tix & 3) != wave) continue;
                    double af[4], bf[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { af[q] = -a[I0 + li][blk + 4 * q + lq]; bf[q] = a[J0 + li][blk + 4 * q + lq]; }
                    pt_f64x4 cf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cf[r] = a[I0 + lq + 4 * r][J0 + li];
#pragma unroll
                    for (int q = 0; q < 4; ++q) cf = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], bf[q], cf, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[I0 + lq + 4 * r][J0 + li] = cf[r];
                }
            __syncthreads();
            if (blk == 0) PT_STAMP(13);
        }
    }
    PT_STAMP(9);
    if (wave != 1) {                                       // (wave 1 is still busy with the last block's inverse)
        const int t3 = (wave == 0 ? 0 : wave - 1) * 64 + lane;
        for (int e = t3; e < NB * NB; e += 192) { const int r = e >> 6, cc = e & 63; Ab[(ri + r) * lda + ri + cc] = (cc <= r) ? a[r][cc] : 0.0; }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(progress + i, i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    PT_STAMP(10);
    }
    if (tid == 0 && !nowait) {
        if (lost && info) info[b] = -1;
        // the workgroup that completes last leaves the counters at zero for the next launch (every poll of a counter precedes the poller's own completion)
        if (__hip_atomic_fetch_add(progress + nbk, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nbk - 1)
            for (int r = 0; r <= nbk; ++r) __hip_atomic_store(progress + r, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    {   // 16 loads per thread, issued back to back before the LDS stores (see potrf_panel_kernel)
        T vl[NB * NB / 256];
#pragma unroll
        for (int it = 0; it < NB * NB / 256; ++it) {
            const int e = tid + it * 256, i = e / NB, m = e % NB;
            T v = (T)0;
            if (i < nb && m < nb) {
                if (m <= i) v = TRANS ? Lkk[(int64_t)(nb - 1 - m) * ldl + (nb - 1 - i)] : Lkk[(int64_t)i * ldl + m];
            } else if (i == m) v = (T)1;
            vl[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NB * NB / 256; ++it) { const int e = tid + it * 256; l[e / NB][e % NB] = vl[it]; }
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

// ---- inverse of every 64x64 diagonal block of L in ONE launch (one wave per block) -----------------------------------------------------
// Blocked: the four 16 x 16 diagonal sub-blocks by substitution (16 lanes per sub-block, one column of the inverse per lane, the column in
// registers), then two merge levels  inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1, C^-1]  as 16 x 16 MFMA products on the LDS image.
// (The un-blocked form -- one column per lane, 64 dependent steps through LDS -- took ~120 us.)
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lq, int r) { return lq + 4 * r; }       // C/D: row = (lane >> 4) + 4 r
};
template <> struct Mfma16<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lq, int r) { return lq * 4 + r; }       // C/D: row = (lane >> 4) * 4 + r
};
// D (16 x 16 at Dst[d0r.., d0c..]) = sgn * A (16 x K16) * B (K16 x 16), A at As[a0r.., a0c..], B at Bs[b0r.., b0c..]; K16 = 16 * kt
template <typename T>
__device__ __forceinline__ void lds_mm16(T (*Dst)[NB + 1], int d0r, int d0c, T (*As)[NB + 1], int a0r, int a0c, T (*Bs)[NB + 1], int b0r, int b0c,
                                         int kt, T sgn, int lane) {
    const int li = lane & 15, lq = lane >> 4;
    typename Mfma16<T>::vec c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = (T)0;
    for (int k = 0; k < 16 * kt; k += 4) c = Mfma16<T>::mma(sgn * As[a0r + li][a0c + k + lq], Bs[b0r + k + lq][b0c + li], c);
#pragma unroll
    for (int r = 0; r < 4; ++r) Dst[d0r + Mfma16<T>::row(lq, r)][d0c + li] = c[r];
}
template <typename T>
__global__ __launch_bounds__(64) void trtri_diag_kernel(const T* __restrict__ L, int64_t ldl, int64_t sL, T* __restrict__ Li, int64_t ldi,
                                                        int64_t sI, int64_t n) {
    __shared__ T l[NB][NB + 1];
    __shared__ T x[NB][NB + 1];
    __shared__ T w[NB][NB + 1];              // products L21 I11 of the merge levels
    const int tid = threadIdx.x, b = blockIdx.y;
    const int64_t k0 = (int64_t)blockIdx.x * NB;
    const int nb = (int)((k0 + NB < n) ? NB : n - k0);
    const T* Lkk = L + (int64_t)b * sL + k0 * ldl + k0;
    T* Ikk = Li + (int64_t)b * sI + k0 * ldi + k0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {       // 2 x 32 loads per lane, each batch issued back to back (see potrf_panel_kernel)
        T vl[NB / 2];
#pragma unroll
        for (int it = 0; it < NB / 2; ++it) {
            const int e = tid + (half * (NB / 2) + it) * 64, i = e / NB, m = e % NB;
            T v = (T)0;
            if (i < nb && m < nb) { if (m <= i) v = Lkk[(int64_t)i * ldl + m]; }
            else if (i == m) v = (T)1;          // identity padding of a ragged last block
            vl[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NB / 2; ++it) {
            const int e = tid + (half * (NB / 2) + it) * 64;
            l[e / NB][e % NB] = vl[it];
            x[e / NB][e % NB] = (T)0;
        }
    }
    __syncthreads();
    {   // 16 x 16 diagonal sub-blocks: lane group g = tid / 16 owns sub-block g, lane c = tid % 16 column c of its inverse
        const int g0 = (tid >> 4) * 16, c = tid & 15;
        T xc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T sacc = (i == c) ? (T)1 : (T)0;
#pragma unroll
            for (int m = 0; m < i; ++m) sacc = fma(-l[g0 + i][g0 + m], (m >= c) ? xc[m] : (T)0, sacc);
            xc[i] = (i >= c) ? sacc / l[g0 + i][g0 + i] : (T)0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) x[g0 + i][g0 + c] = xc[i];
    }
    __syncthreads();
    // merge 16 -> 32: pairs (0,1) and (2,3):  X21 = -I22 (L21 I11)
    for (int p = 0; p < 2; ++p) {
        const int o = 32 * p;
        lds_mm16<T>(w, o + 16, o, l, o + 16, o, x, o, o, 1, (T)1, tid);                 // W = L21 I11
    }
    __syncthreads();
    for (int p = 0; p < 2; ++p) {
        const int o = 32 * p;
        lds_mm16<T>(x, o + 16, o, x, o + 16, o + 16, w, o + 16, o, 1, (T)-1, tid);      // X21 = -I22 W
    }
    __syncthreads();
    // merge 32 -> 64: L21 = l[32:64][0:32], I11 = x[0:32][0:32], I22 = x[32:64][32:64] (both lower triangular, explicit zeros above)
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) lds_mm16<T>(w, 32 + 16 * ti, 16 * tj, l, 32 + 16 * ti, 0, x, 0, 16 * tj, 2, (T)1, tid);
    __syncthreads();
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) lds_mm16<T>(x, 32 + 16 * ti, 16 * tj, x, 32 + 16 * ti, 32, w, 32, 16 * tj, 2, (T)-1, tid);
    __syncthreads();
    for (int e = tid; e < nb * nb; e += 64) {
        const int i = e / nb, m = e % nb;
        Ikk[(int64_t)i * ldi + m] = x[i][m];
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

// ---- the block rows BELOW an outer panel (r03) ------------------------------------------------------------------------------------------
// After the panel's own block rows are factored (potrf_tiles_kernel, the latency chain), every row below solves  L[i][0..npt) L11^T = A[i][0..npt)
// against the FINISHED diagonal block L11.  potrf_tiles_kernel did that one 64 x 64 tile at a time, left-looking, the whole workgroup on one
// tile: ~37 us per tile, 0.3 ms per outer panel at n = 8192 although a block row's matrix work is 25 us (these launches were 2.5 of
// potrf(8192)'s 7.3 ms, and only rows / 64 <= 120 of the 256 CUs had a workgroup).  Here a workgroup of EIGHT waves keeps the block row's eight
// tiles in registers, one per wave (64 x 64 float64 = 128 VGPRs in the accumulator layout), and goes RIGHT-looking:
//   step k: the owner of tile k puts it into LDS (two buffers, alternating);  waves 0..3 solve  X L[k][k]^T = T  in place, 16 rows each, with
//   the published inverses of L[k][k]'s four 16 x 16 diagonal blocks (the substitution of potrf_tiles_kernel);  X goes to global memory (it is
//   L[i][k]);  every wave j > k updates its tile in place,  A[i][j] -= X L[j][k]^T,  the L[j][k] operand straight from global memory (L11 is
//   2 MB: L2-resident, every workgroup reads the same tiles) three 16-column groups ahead of the MFMAs that use them.
// Two barriers per step; the critical chain of a block row is npt x (solve + one tile update by one wave, 256 MFMAs).
// (Measured and dropped: this launch NEXT TO the panel's chain on a second auxiliary stream, following the chain's progress counters column
//  by column -- it then ends 17 us after the chain, but the chain itself takes 0.30 - 0.36 instead of 0.19 ms (the followers' acquire fences
//  and tile reads share its L2) and 120 mostly waiting workgroups hold CUs the look-ahead GEMM wants: potrf(8192) 7.18 vs 6.72 ms.)
// MFMA conventions as potrf_tiles_kernel: A operand lane (li, lq) = P[m = li][k = lq], B operand = Q[n = li][k = lq], accumulator register r of
// lane (li, lq) = element (row lq + 4 r, column li).
// RH (r06): rows per workgroup.  A block row's critical chain is npt x (solve + ONE wave's tile update): at RH = 64 that update is 256 float64
// MFMAs of 64 cycles each, 7.8 us, i.e. >= 62 us per outer panel on <= 120 of the 256 CUs.  RH = 32 / 16 halve / quarter the chain and
// double / quadruple the workgroup count (each re-reads the L2-resident diagonal block L11: 1.2 MB per workgroup).
static_assert(NBO / NB <= 8, "potrf_rows_kernel: one wave per tile of the panel, eight waves");
template <int RH>
__global__ __launch_bounds__(512) void potrf_rows_kernel(double* __restrict__ A, int64_t lda, int64_t sA, int64_t c0, int npt, int row0,
                                                         const double* __restrict__ inv_all) {
    constexpr int XR = RH / 16;
    __shared__ double Tb[2][RH][NB + 1];
    __shared__ double a[NB][NB + 1];          // L[k][k]
    __shared__ double minv[4][16][17];        // the inverses of its four 16 x 16 diagonal blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y;
    double* Ab = A + (int64_t)b * sA;
    const double* invs = inv_all + (int64_t)b * npt * 1024;
    const int64_t ri = c0 + (int64_t)row0 * NB + (int64_t)blockIdx.x * RH;      // (row0 in units of NB rows)
    const bool own = wave < npt;
    const int64_t rjw = c0 + (int64_t)wave * NB;
    pt_f64x4 acc[XR][4];
    if (own) {
#pragma unroll
        for (int x = 0; x < XR; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[x][y][r] = Ab[(ri + 16 * x + lq + 4 * r) * lda + rjw + 16 * y + li];
    }
    for (int k = 0; k < npt; ++k) {
        const int64_t rk = c0 + (int64_t)k * NB;
        double (*T)[NB + 1] = Tb[k & 1];
        // operands that do not depend on this step's tile: requested in front of the barriers
        double lv[8];                           // L[k][k], element e = tid + 512 u <-> (row e / 64, column e % 64); lower triangle
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = tid + 512 * u, rr = e >> 6, cc = e & 63;
            lv[u] = (cc <= rr) ? Ab[(rk + rr) * lda + rk + cc] : 0.0;
        }
        const double mv0 = invs[(int64_t)k * 1024 + tid], mv1 = invs[(int64_t)k * 1024 + 512 + tid];
        const bool upd = own && wave > k;
        const double* lj = Ab + (rjw + li) * lda + rk + lq;            // L[j = wave][k][n = li + 16 y][c = lq + 4 q]
        // the update's B operand in groups of four k steps (16 columns of L[j][k]), three groups ahead of the MFMAs that use them
        auto lgroup = [&](int g, double (&d)[4]) {           // group g = 4 y + qq: rows 16 y + li, columns 16 qq + 4 t + lq
            const double* p = lj + (int64_t)16 * (g >> 2) * lda + 16 * (g & 3);
#pragma unroll
            for (int t = 0; t < 4; ++t) d[t] = p[4 * t];
        };
        double b0[4], b1[4], b2[4], b3[4];
        if (upd) { lgroup(0, b0); lgroup(1, b1); lgroup(2, b2); }
        if (wave == k) {
#pragma unroll
            for (int x = 0; x < XR; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[16 * x + lq + 4 * r][16 * y + li] = acc[x][y][r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = tid + 512 * u; a[e >> 6][e & 63] = lv[u]; }
        minv[tid >> 8][(tid >> 4) & 15][tid & 15] = mv0;
        minv[2 + (tid >> 8)][(tid >> 4) & 15][tid & 15] = mv1;
        __syncthreads();
        if (wave < XR) {
            // X L[k][k]^T = T, 16 rows per wave, in registers (accumulator layout of the TRANSPOSED tile: lane (li, lq), register r = element
            // (row li, column lq + 4 r) of a 16 x 16 block): X_b = Y_b M_b^T with the published inverse M_b (4 MFMAs), then
            // Y_b'' -= X_b L_b''b^T for the blocks to the right -- as potrf_tiles_kernel
            pt_f64x4 mreg[4], tacc[4];
            const int rw = wave * 16;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { mreg[bb][r] = minv[bb][li][4 * r + lq]; tacc[bb][r] = T[rw + li][16 * bb + lq + 4 * r]; }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                pt_f64x4 xs = pt_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) xs = __builtin_amdgcn_mfma_f64_16x16x4f64(mreg[bb][r], tacc[bb][r], xs, 0, 0, 0);
#pragma unroll
                for (int b2i = bb + 1; b2i < 4; ++b2i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tacc[b2i] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[16 * b2i + li][16 * bb + 4 * r + lq], xs[r], tacc[b2i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) T[rw + li][16 * bb + lq + 4 * r] = xs[r];
            }
        }
        __syncthreads();
        {   // L[i][k] = X to global memory, coalesced
#pragma unroll
            for (int u = 0; u < RH / 8; ++u) { const int e = tid + 512 * u, rr = e >> 6, cc = e & 63; Ab[(ri + rr) * lda + rk + cc] = T[rr][cc]; }
        }
        if (upd) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
#pragma unroll 1
                for (int qq = 0; qq < 4; ++qq) {
                    const int g = 4 * y + qq;
                    if (g + 3 < 16) lgroup(g + 3, b3);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int x = 0; x < XR; ++x)
                            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[16 * x + li][16 * qq + 4 * t + lq], b0[t], acc[x][y], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { b0[t] = b1[t]; b1[t] = b2[t]; b2[t] = b3[t]; }
                }
            }
        }
    }
}

template <typename T>
int trtri_typed(mxf_ctx* h, int dtype, int S, int64_t n, const T* L, int64_t ldl, int64_t sL, T* Li, int64_t ldi, int64_t sI, hipStream_t st);
template <typename T>
__global__ void zero_block_kernel(T* __restrict__ P, int64_t rows, int64_t cols, int64_t ld, int64_t stride);

// Row block i (rows [c0, pe)) of L^-1 from the finished leading part of the factor (r05; the merge step of trtri_typed with a first block of c0 rows
// and a second of pe - c0):  I_ii = inv(L_ii),  X = -I_ii (L[i, :c0] I[:c0, :c0]),  the temporary (L[i, :c0] I[:c0, :c0])^T in the upper mirror block.
// Needs L[:pe, :pe] only, i.e. it can run as soon as the outer panel ending at pe has been factored -- next to the rest of the factorisation.
template <typename T>
int trtri_row_block(mxf_ctx* h, int dtype, int64_t c0, int64_t pe, const T* L, int64_t ldl, T* Li, int64_t ldi, hipStream_t st) {
    const int64_t b2 = pe - c0;
    int rc = trtri_typed<T>(h, dtype, 1, b2, L + c0 * (ldl + 1), ldl, 0, Li + c0 * (ldi + 1), ldi, 0, st);
    if (rc || c0 == 0) return rc;
    static const int res = (int)MXF_KNOB("MXF_POTRF_EAGER_RES", 0);      // CUs these products leave to the factorisation's chain (probe knob)
    rc = mxf_gemm_internal(h, dtype, 1, 1, c0, b2, c0, 1.0, Li, ldi, 0, L + c0 * ldl, ldl, 0, 0.0, Li + c0, ldi, 0, 1, 0, st, res, 1);
    if (rc) return rc;
    rc = mxf_gemm_internal(h, dtype, 0, 1, b2, c0, b2, -1.0, Li + c0 * (ldi + 1), ldi, 0, Li + c0, ldi, 0, 0.0, Li + c0 * ldi, ldi, 0, 1, 0, st, res, 2);
    if (rc) return rc;
    hipLaunchKernelGGL((zero_block_kernel<T>), dim3((unsigned)((c0 * b2 + 255) / 256 > 1024 ? 1024 : (c0 * b2 + 255) / 256), 1), dim3(256), 0, st, Li + c0, c0, b2, ldi, (int64_t)0);
    return 0;
}
// The same row block in its three dependent pieces (r06), so that each starts as soon as ITS inputs exist:
//   p1: (L[i, :c0] I[:c0, :c0])^T into the upper mirror block -- needs the rows below the panel ending at c0 and the finished leading inverse,
//       NOT the block's own panels (the bulk of the row block's work: 2 b2 c0^2 / 2 flops);
//   the inverse of the diagonal block L[i, i] -- needs the block's own panels only (latency-bound small launches: a stream of its own);
//   p2: X = -I_ii p1^T, then the mirror block zeroed again.
template <typename T>
int trtri_row_block_p1(mxf_ctx* h, int dtype, int64_t c0, int64_t pe, const T* L, int64_t ldl, T* Li, int64_t ldi, hipStream_t st) {
    return mxf_gemm_internal(h, dtype, 1, 1, c0, pe - c0, c0, 1.0, Li, ldi, 0, L + c0 * ldl, ldl, 0, 0.0, Li + c0, ldi, 0, 1, 0, st, 0, 1);
}
template <typename T>
int trtri_row_block_p2(mxf_ctx* h, int dtype, int64_t c0, int64_t pe, T* Li, int64_t ldi, hipStream_t st) {
    const int64_t b2 = pe - c0;
    int rc = mxf_gemm_internal(h, dtype, 0, 1, b2, c0, b2, -1.0, Li + c0 * (ldi + 1), ldi, 0, Li + c0, ldi, 0, 0.0, Li + c0 * ldi, ldi, 0, 1, 0, st, 0, 2);
    if (rc) return rc;
    hipLaunchKernelGGL((zero_block_kernel<T>), dim3((unsigned)((c0 * b2 + 255) / 256 > 1024 ? 1024 : (c0 * b2 + 255) / 256), 1), dim3(256), 0, st, Li + c0, c0, b2, ldi, (int64_t)0);
    return 0;
}

// Ie != nullptr (float64, one matrix, the per-panel tile form with look-ahead): L^-1 is formed into Ie row block by row block on a third stream while
// the factorisation goes on -- potrf(8192) is bound by its serial path (16 x (head update + chain + rows below)), the chip is ~40 % idle under it, and
// trtri(8192) afterwards took 3.9 ms of a 15 ms MAP step.  *eager_done tells the caller whether Ie was filled (else: call trtri afterwards).
template <typename T>
int potrf_typed(mxf_ctx* h, int dtype, int S, int64_t n, T* A, int64_t lda, int64_t sA, int* info, hipStream_t st, bool zero_upper, bool zero_info,
                T* Ie = nullptr, int64_t ldie = 0, bool* eager_done = nullptr, T* Kacc = nullptr, int64_t ldk = 0, double kcoef = 0.0,
                bool* kacc_done = nullptr) {
    if (eager_done) *eager_done = false;
    if (kacc_done) *kacc_done = false;
    if (info && zero_info) MXF_HIP(h, hipMemsetAsync(info, 0, sizeof(int) * S, st));
    // Look-ahead (n >= 2048): the trailing update after an outer panel is split into the part that touches the NEXT outer panel's columns
    // (on the caller's stream, so that panel's latency-bound factorisation starts right away) and the rest (on an auxiliary stream, next to
    // that factorisation).  At n = 8192 the 128 panel steps (85 us each) otherwise serialise with 3.7 ms of trailing GEMMs.
    static const int tiles_env = MXF_KNOB("MXF_POTRF_TILES", 1);
    // tiles_env: 0 = launch-per-panel form everywhere, 1 = the tile kernel (one launch up to MXF_POTRF_ONE_MAX = 512, per outer panel beyond), 2 = only its one-launch form
    // The tile kernel's workgroups hand tiles to each other through progress counters: every workgroup of a launch must be RESIDENT at once
    // (one per CU: 256 threads at one wave per SIMD, ~76 KB of LDS), or a waiting workgroup could hold the CU its producer needs.  The grid
    // (block rows x batch) is therefore bounded by the CU count of the device; larger problems take the launch-per-panel form below.
    static const int ncu = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 64; return v; }();
    const bool tiles_ok = sizeof(T) == 8 && tiles_env && n % NB == 0 && n >= 2 * NB && S <= 64;
    if constexpr (sizeof(T) == 8) {
        static const int one_max = MXF_KNOB("MXF_POTRF_ONE_MAX", 512);
        if (tiles_ok && n <= one_max && (int64_t)(n / NB) * S <= ncu) {    // (n = 2048 as ONE left-looking launch: 2.6 ms vs 1.9 -- the last block rows carry 32 i^2 columns of products each)
            const unsigned nbk = (unsigned)(n / NB);
            int* progress = mxf_flags(h, (nbk + 1) * (unsigned)S);
            if (!progress) MXF_FAIL(h, -4, "mxf_potrf: cannot allocate the workgroup hand-off counters");
            double* pinv = mxf_potrf_inv(h, (size_t)nbk * S * 1024);
            if (!pinv) MXF_FAIL(h, -4, "mxf_potrf: cannot allocate the inverse-block scratch");
            hipLaunchKernelGGL(potrf_tiles_kernel, dim3(nbk, (unsigned)S), dim3(256), 0, st, A, lda, sA, (int64_t)0, (int)nbk, info, progress, pinv, 0, 0);
            if (zero_upper) hipLaunchKernelGGL((zero_upper_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)n, S), dim3(256), 0, st, A, n, lda, sA);
            MXF_LAUNCH_CHECK(h);
            return 0;
        }
    }
    const bool panel_tiles = tiles_ok && tiles_env == 1 && (int64_t)(n / NB) * S <= ncu;      // every block row's workgroup must be resident at once
    static const int look_env = MXF_KNOB("MXF_POTRF_LOOKAHEAD", 2);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    const bool look = look_env && n >= 2048 && cap == hipStreamCaptureStatusNone && mxf_potrf_aux_init(h);
    // MXF_POTRF_CUMASK (common.h): CU-masked bulk streams are blocking streams -- the chain moves to a non-blocking stream of its own
    hipStream_t st_user = st;
    const bool own_chain = look && (h->potrf_masked || h->potrf_chain_always) && h->potrf_chain;
    if (own_chain) {
        MXF_HIP(h, hipEventRecord(h->ev_pk, st_user));
        MXF_HIP(h, hipStreamWaitEvent(h->potrf_chain, h->ev_pk, 0));
        st = h->potrf_chain;
    }
    hipStream_t ax = look ? h->potrf_aux : st;
    bool pending_b = false, pending_h = false, pending_r = false;
    // r06, measured and NOT kept -- probe knob MXF_POTRF_ROWS2=1: only the NEXT panel's eight block rows of the rows below
    // an outer panel are on the serial path (its diagonal block's head update reads them), so the rows further down are solved on a stream
    // of their own next to that head update and the next chain, and the auxiliary stream's products -- the only readers of those rows --
    // wait for them.  Correct (tests/test_gpu_linalg.py passes either way), but same box, alternating: potrf(8192) 6.65 / 6.67 ms with it,
    // 6.57 / 6.60 without; exact-GP MAP step 14.64-14.67 against 14.52-14.53 ms -- the hundred far-row workgroups now run NEXT TO the next
    // panel's latency-bound chain and head update instead of before them, and the chain pays more than the shorter path saves (the same
    // outcome as r03's rows kernel following the chain's progress counters and r05's per-panel eager inverse).
    static const int rows2_env = MXF_KNOB("MXF_POTRF_ROWS2", 0);
    // row blocks of FOUR outer panels (2048 rows), from four row blocks on.  Measured at n = 8192 (MAP step of the exact GP, two alternating rounds,
    // profiles/r05_potrf_eager_inverse_ab.txt): off 15.0 ms; every panel 18.1 (208 more launches, and products of a few tiles each that hold CUs the
    // chain's tile workgroups are waiting for); every second 14.6-14.7; every fourth 14.5; two halves 15.0; leaving the products 64 / 128 CUs less changes nothing
    // r06: with the row blocks in split pieces (MXF_POTRF_EAGER_SPLIT below) blocks of TWO panels are best: 14.10-14.15 ms against 14.42-14.44
    // (four), 15.0 (three: ragged last block), 14.72 for the r05 form
    static const int eager_env = MXF_KNOB("MXF_POTRF_EAGER_INV", 2);
    // r06 (session 4), measured and NOT kept -- probe knob MXF_POTRF_KACC=1: kcoef L^-T L^-1 = sum over row blocks b of Linv[b, :]^T Linv[b, :],
    // each share accumulated into the caller's zeroed buffer (lower triangle) on a stream of its own as soon as row block b of the eager inverse
    // is final -- 2.6 ms of products moved under the factorisation, leaving 1.2 ms (the last block's share) instead of the 2.9 ms product behind it.
    // Correct (the exact-GP tests pass with it), but the MAP step at n = 8192 goes 13.82-13.87 -> 15.73-15.84 ms (same box, alternating,
    // tests/probes/r06_kacc_ab*.sh): the chain pays ~4 ms for the 1.7 ms the tail saves.  What the chain suffers from is not a lack of free
    // CUs: with the bulk streams CU-masked so that 2 / 4 / 8 CUs of every XCD stay free for it (MXF_POTRF_CUMASK, common.h; the masked streams
    // are blocking streams, so the chain then runs on a non-blocking stream of its own) the step takes 18.5-18.7 ms (22.3-24.0 with the
    // accumulation), and the chain alone on a most-urgent stream (MXF_POTRF_CHAIN_PRIO=1) 16.8 ms; GPU_MAX_HW_QUEUES 4 / 8 changes nothing.
    // Its tile hand-offs go through L2 / the fabric, and that is what the products next to it load.
    static const int kacc_env = MXF_KNOB("MXF_POTRF_KACC", 0);
    const bool eager = Ie != nullptr && eager_env && sizeof(T) == 8 && S == 1 && panel_tiles && look && n % NBO == 0 && n >= 16 * NBO &&
                       n >= 4 * (eager_env >= 100 ? n / 8 : (int64_t)eager_env * NBO);
    static const int esplit_env0 = MXF_KNOB("MXF_POTRF_EAGER_SPLIT", 1);
    const bool kacc = eager && Kacc != nullptr && kacc_env && esplit_env0 && !rows2_env && h->potrf_acc;
    if (kacc) {     // (the buffer is zeroed on the accumulation stream, behind whatever the caller's stream did with it before)
        MXF_HIP(h, hipEventRecord(h->ev_pq, st));
        MXF_HIP(h, hipStreamWaitEvent(h->potrf_acc, h->ev_pq, 0));
        MXF_HIP(h, hipMemsetAsync(Kacc, 0, sizeof(T) * (size_t)n * (size_t)ldk, h->potrf_acc));
    }
    static const int split_rows_g = MXF_KNOB("MXF_POTRF_SPLIT_ROWS", 64);
    static const int rows_env_g = MXF_KNOB("MXF_POTRF_ROWS_KERNEL", 1);     // 0: the rows below through potrf_tiles_kernel (r02)
    static const int head_split_env = MXF_KNOB("MXF_POTRF_HEAD_SPLIT", 1);  // 1: the next panel's rows-below head update on the auxiliary stream
    // does the outer panel at c0 take the split form (chain launch + rows-below launch)?
    auto is_split = [&](int64_t c0_) {
        const int64_t pe_ = (c0_ + NBO < n) ? c0_ + NBO : n;
        const int64_t nbr_ = (n - c0_) / NB, npt_ = (pe_ - c0_) / NB;
        return panel_tiles && split_rows_g > 0 && nbr_ - npt_ >= split_rows_g;
    };
    for (int64_t c0 = 0; c0 < n; c0 += NBO) {
        const int64_t pe = (c0 + NBO < n) ? c0 + NBO : n;   // panel end
        //  the auxiliary stream's head update of THIS panel's rows-below is consumed here unconditionally, whatever form the
        // panel takes: the dependency must not hinge on is_split() implying the tile path.  In the split tile form the wait is deferred to
        // just in front of the rows kernel (the chain launch reads the diagonal block only) -- `defer` below.
        const bool defer_h = pending_h && sizeof(T) == 8 && panel_tiles && is_split(c0);
        if (pending_h && !defer_h) { MXF_HIP(h, hipStreamWaitEvent(st, h->ev_ph, 0)); pending_h = false; }
        if constexpr (sizeof(T) == 8) {
            if (panel_tiles) {       // the whole outer panel in ONE launch (8 dependent panel steps + 7 left-looking GEMMs before)
                const unsigned nbr = (unsigned)((n - c0) / NB), npt = (unsigned)((pe - c0) / NB);
                // Many block rows below the panel: two launches -- the panel's own block rows (the latency chain, npt workgroups), then the
                // rows below against the finished diagonal block, nothing to wait for (~90 us of MFMA work each).  In one launch those rows
                // sit resident and mostly idle for the whole chain, one CU each, and the look-ahead GEMM next to them (whose 133 KB of LDS
                // cannot share a CU with a tile workgroup) runs on what is left.
                const bool split = is_split(c0);
                const unsigned na = split ? npt : nbr;
                int* progress = mxf_flags(h, (na + 1) * (unsigned)S);
                if (!progress) MXF_FAIL(h, -4, "mxf_potrf: cannot allocate the workgroup hand-off counters");
                double* pinv = mxf_potrf_inv(h, (size_t)npt * S * 1024);
                if (!pinv) MXF_FAIL(h, -4, "mxf_potrf: cannot allocate the inverse-block scratch");
                hipLaunchKernelGGL(potrf_tiles_kernel, dim3(na, (unsigned)S), dim3(256), 0, st, A, lda, sA, c0, (int)npt, info, progress, pinv, 0, 0);
                const int rows_env = rows_env_g;
                // (the rows below this panel's diagonal block were updated on the auxiliary stream, next to the chain above)
                const bool had_h = pending_h;
                if (pending_h) { MXF_HIP(h, hipStreamWaitEvent(st, h->ev_ph, 0)); pending_h = false; }
                // nblk block rows of NB rows from block row row0 of the panel on: workgroups of RH rows each (MXF_POTRF_ROWS_RH, see the kernel)
                auto launch_rows = [&](unsigned nblk, unsigned row0, hipStream_t s_) {
                    static const int rh_env = MXF_KNOB("MXF_POTRF_ROWS_RH", 32);
                    if (rh_env == 16) hipLaunchKernelGGL(potrf_rows_kernel<16>, dim3(nblk * 4, (unsigned)S), dim3(512), 0, s_, A, lda, sA, c0, (int)npt, (int)row0, (const double*)pinv);
                    else if (rh_env == 32) hipLaunchKernelGGL(potrf_rows_kernel<32>, dim3(nblk * 2, (unsigned)S), dim3(512), 0, s_, A, lda, sA, c0, (int)npt, (int)row0, (const double*)pinv);
                    else hipLaunchKernelGGL(potrf_rows_kernel<64>, dim3(nblk, (unsigned)S), dim3(512), 0, s_, A, lda, sA, c0, (int)npt, (int)row0, (const double*)pinv);
                };
                if (split && rows_env) {       // r03: the rows below right-looking from registers (potrf_rows_kernel)
                    const unsigned nbel = nbr - npt, nfirst = (unsigned)(NBO / NB) < nbel ? (unsigned)(NBO / NB) : nbel;
                    if (rows2_env && look && head_split_env && nbel >= nfirst + 16 && pe + NBO < n) {
                        if (had_h) MXF_HIP(h, hipStreamWaitEvent(h->potrf_rows, h->ev_ph, 0));      // (their columns' head update, auxiliary stream)
                        MXF_HIP(h, hipEventRecord(h->ev_pc, st));                                   // the chain of this panel (and all before it)
                        MXF_HIP(h, hipStreamWaitEvent(h->potrf_rows, h->ev_pc, 0));
                        launch_rows(nfirst, npt, st);
                        launch_rows(nbel - nfirst, npt + nfirst, h->potrf_rows);
                        MXF_HIP(h, hipEventRecord(h->ev_rb, h->potrf_rows));
                        pending_r = true;
                    } else
                    launch_rows(nbr - npt, npt, st);
                } else if (split)
                    hipLaunchKernelGGL(potrf_tiles_kernel, dim3(nbr - npt, (unsigned)S), dim3(256), 0, st, A, lda, sA, c0, (int)npt, info, progress, pinv, (int)npt, 1);
            }
        }
        // row block [rb0, pe) of L is final once the panel ending at pe has been factored (the rows above it in these columns are zero).  Row blocks
        // of eager_rb outer panels (probe knob MXF_POTRF_EAGER_INV: 1 = every panel, 2 = every second, ..., 100 = two halves)
        const int64_t rbw = eager_env >= 100 ? n / 2 : (int64_t)eager_env * NBO;
        // r06 probe knob MXF_POTRF_EAGER_TAIL = t > 0: behind the first 3/4 of the rows the row blocks shrink to t outer panels -- the last
        // block is what stays exposed behind the factorisation (2.5 ms of the 14.6 ms MAP step at n = 8192 with blocks of four panels), and
        // the late panels are chain-bound, i.e. the chip is mostly idle next to them
        static const int tail_env = MXF_KNOB("MXF_POTRF_EAGER_TAIL", 0);
        const int64_t tail0 = (n * 3 / 4) / rbw * rbw, tbw = (int64_t)(tail_env > 0 ? tail_env : 1) * NBO;
        const bool in_tail = tail_env > 0 && eager_env < 100 && tbw < rbw && pe > tail0;
        const bool fire = in_tail ? ((pe - tail0) % tbw == 0 || pe == n) : (pe % rbw == 0 || pe == n);
        // r06, MXF_POTRF_EAGER_SPLIT (default 1): the row block's pieces separately (trtri_row_block_p1 / _p2).  When block b's panels end, its
        // diagonal inverse goes to the third auxiliary stream, its p2 follows on the inverse stream, and p1 of block b + 1 -- whose inputs are
        // complete at this point, four panels before that block is factored -- is queued right behind.  What is left behind the factorisation
        // is the last block's diagonal inverse and p2 instead of its whole row block.
        static const int esplit_env = MXF_KNOB("MXF_POTRF_EAGER_SPLIT", 1);
        const bool esplit = esplit_env && !rows2_env;
        auto fires_at = [&](int64_t pe_) {
            const bool it_ = tail_env > 0 && eager_env < 100 && tbw < rbw && pe_ > tail0;
            return it_ ? ((pe_ - tail0) % tbw == 0 || pe_ == n) : (pe_ % rbw == 0 || pe_ == n);
        };
        if (eager && fire) {
            const int64_t rb0 = in_tail ? tail0 + (pe - tail0 - 1) / tbw * tbw : (pe - 1) / rbw * rbw;
            MXF_HIP(h, hipEventRecord(h->ev_pi, st));
            if constexpr (sizeof(T) == 8) {
                if (esplit) {
                    hipStream_t qd = h->potrf_rows, qp = h->potrf_inv;
                    MXF_HIP(h, hipStreamWaitEvent(qd, h->ev_pi, 0));
                    int rc = trtri_typed<T>(h, dtype, 1, pe - rb0, A + rb0 * (lda + 1), lda, 0, Ie + rb0 * (ldie + 1), ldie, 0, qd);
                    if (rc) return rc;
                    MXF_HIP(h, hipEventRecord(h->ev_pc, qd));
                    MXF_HIP(h, hipStreamWaitEvent(qp, h->ev_pc, 0));       // (block 0: p1 of block 1 reads this inverse)
                    if (rb0 > 0) { rc = trtri_row_block_p2<T>(h, dtype, rb0, pe, Ie, ldie, qp); if (rc) return rc; }
                    if (kacc) {
                        // rows [rb0, pe) of L^-1 are final (columns < pe; zero beyond): their share of kcoef L^-T L^-1, on a stream of its own so
                        // that the next block's p1 is not held up behind it (the shares update the same tiles: one stream keeps them in order)
                        MXF_HIP(h, hipEventRecord(h->ev_pq, qp));
                        MXF_HIP(h, hipStreamWaitEvent(h->potrf_acc, h->ev_pq, 0));
                        rc = mxf_gemm_internal(h, dtype, 1, 0, pe, pe, pe - rb0, kcoef, Ie + rb0 * ldie, ldie, 0, Ie + rb0 * ldie, ldie, 0, 1.0, Kacc, ldk, 0,
                                               1, 1, h->potrf_acc, 0, 0);
                        if (rc) return rc;
                    }
                    if (pe < n) {
                        MXF_HIP(h, hipStreamWaitEvent(qp, h->ev_pi, 0));   // the rows below the panel that just ended
                        int64_t ne = pe + NBO;                           // the end of the next row block = the next firing point
                        while (ne < n && !fires_at(ne)) ne += NBO;
                        if (ne > n) ne = n;
                        rc = trtri_row_block_p1<T>(h, dtype, pe, ne, A, lda, Ie, ldie, qp);
                        if (rc) return rc;
                    }
                } else {
                    MXF_HIP(h, hipStreamWaitEvent(h->potrf_inv, h->ev_pi, 0));
                    int rc = trtri_row_block<T>(h, dtype, rb0, pe, A, lda, Ie, ldie, h->potrf_inv);
                    if (rc) return rc;
                }
            }
        }
        for (int64_t j0 = c0; j0 < pe && !panel_tiles; j0 += NB) {
            const int nb = (int)((j0 + NB < n) ? NB : n - j0);
            if (j0 > c0) {   // left-looking update of block column j0 with the panel's previous block columns
                int rc = mxf_gemm_internal(h, dtype, 0, 1, n - j0, nb, j0 - c0, -1.0, A + j0 * lda + c0, lda, sA,
                                           A + j0 * lda + c0, lda, sA, 1.0, A + j0 * lda + j0, lda, sA, S, 0, st);
                if (rc) return rc;
            }
            const int64_t below = n - (j0 + nb);
            int* arrived = mxf_flags(h, (unsigned)S);
            if (!arrived) MXF_FAIL(h, -4, "mxf_potrf: cannot allocate the workgroup hand-off counters");
            hipLaunchKernelGGL((potrf_panel_kernel<T>), dim3((unsigned)(1 + (below + 127) / 128), S), dim3(128), 0, st, A, lda, sA, j0, nb, n, info,
                               arrived);
        }
        if (pe < n) {   // trailing update, lower blocks only: A22 -= L21 L21^T with K = panel width
            const int64_t pe2 = (pe + NBO < n) ? pe + NBO : n, K = pe - c0;
            if (pending_b) { MXF_HIP(h, hipStreamWaitEvent(st, h->ev_pb, 0)); pending_b = false; }   // the previous rest-update touched these columns
            // the far rows of this panel (potrf_rows stream): every product below except the next diagonal block's head update reads them
            const bool far_rows = pending_r;
            if (far_rows) MXF_HIP(h, hipStreamWaitEvent(ax, h->ev_rb, 0));
            pending_r = false;
            if (!look || pe2 >= n) {
                if (far_rows) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_rb, 0));
                int rc = mxf_gemm_internal(h, dtype, 0, 1, n - pe, n - pe, K, -1.0, A + pe * lda + c0, lda, sA,
                                           A + pe * lda + c0, lda, sA, 1.0, A + pe * lda + pe, lda, sA, S, 1, st);
                if (rc) return rc;
            } else {
                // r03: when the NEXT panel takes the split form, only its diagonal block (what its chain launch reads) is updated on the caller's
                // stream; the rows below it -- read by potrf_rows_kernel only, 0.2 ms later -- are updated on the auxiliary stream next to that
                // chain (0.09 ms per panel off the serial path at n = 8192)
                const bool head_aux = head_split_env && sizeof(T) == 8 && rows_env_g && is_split(pe);
                if (look_env != 2 || head_aux) MXF_HIP(h, hipEventRecord(h->ev_pa, st));  // the panel's columns (L21) are final
                // next outer panel's columns: its diagonal block (lower) and the rows below it
                int rc = mxf_gemm_internal(h, dtype, 0, 1, pe2 - pe, pe2 - pe, K, -1.0, A + pe * lda + c0, lda, sA,
                                           A + pe * lda + c0, lda, sA, 1.0, A + pe * lda + pe, lda, sA, S, 1, st);
                if (rc) return rc;
                if (head_aux) {
                    MXF_HIP(h, hipStreamWaitEvent(ax, h->ev_pa, 0));
                    rc = mxf_gemm_internal(h, dtype, 0, 1, n - pe2, pe2 - pe, K, -1.0, A + pe2 * lda + c0, lda, sA,
                                           A + pe * lda + c0, lda, sA, 1.0, A + pe2 * lda + pe, lda, sA, S, 0, ax);
                    if (rc) return rc;
                    MXF_HIP(h, hipEventRecord(h->ev_ph, ax));
                    pending_h = true;
                } else {
                    if (far_rows) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_rb, 0));
                    rc = mxf_gemm_internal(h, dtype, 0, 1, n - pe2, pe2 - pe, K, -1.0, A + pe2 * lda + c0, lda, sA,
                                           A + pe * lda + c0, lda, sA, 1.0, A + pe2 * lda + pe, lda, sA, S, 0, st);
                    if (rc) return rc;
                    if (look_env == 2) MXF_HIP(h, hipEventRecord(h->ev_pa, st));  // (2: the rest-update only starts once the head products are done)
                    MXF_HIP(h, hipStreamWaitEvent(ax, h->ev_pa, 0));
                }
                // the rest on the auxiliary stream, next to the next panel's factorisation
                rc = mxf_gemm_internal(h, dtype, 0, 1, n - pe2, n - pe2, K, -1.0, A + pe2 * lda + c0, lda, sA,
                                       A + pe2 * lda + c0, lda, sA, 1.0, A + pe2 * lda + pe2, lda, sA, S, 1, ax);
                if (rc) return rc;
                MXF_HIP(h, hipEventRecord(h->ev_pb, ax));
                pending_b = true;
            }
        }
    }
    if (eager) {
        MXF_HIP(h, hipEventRecord(h->ev_pj, h->potrf_inv));
        MXF_HIP(h, hipStreamWaitEvent(st, h->ev_pj, 0));
        if (eager_done) *eager_done = true;
    }
    if (kacc) {
        MXF_HIP(h, hipEventRecord(h->ev_pz, h->potrf_acc));
        MXF_HIP(h, hipStreamWaitEvent(st, h->ev_pz, 0));
        if (kacc_done) *kacc_done = true;
    }
    if (pending_r) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_rb, 0));
    if (pending_b) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_pb, 0));
    if (pending_h) MXF_HIP(h, hipStreamWaitEvent(st, h->ev_ph, 0));      // (never pending here today: the last panel has no successor; kept so that the caller's stream always joins the auxiliary one)
    if (n > 1 && zero_upper) {      // (internal callers that only ever read the lower triangle skip this pass)
        if (n > 65535) MXF_FAIL(h, -3, "mxf_potrf: n too large");
        hipLaunchKernelGGL((zero_upper_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)n, S), dim3(256), 0, st, A, n, lda, sA);
    }
    if (own_chain) {
        MXF_HIP(h, hipEventRecord(h->ev_pk, st));
        MXF_HIP(h, hipStreamWaitEvent(st_user, h->ev_pk, 0));
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

int mxf_potrf_internal(mxf_ctx* h, int dtype, int S, int64_t n, void* A, int64_t lda, int64_t sA, int* info, hipStream_t st, bool zero_upper, bool zero_info,
                       void* Linv_eager, int64_t ldie, bool* eager_done, void* Kacc, int64_t ldk, double kcoef, bool* kacc_done) {
    if (eager_done) *eager_done = false;
    if (kacc_done) *kacc_done = false;
    if (n <= 0 || S <= 0) return 0;
    if (dtype == MXF_F32) return potrf_typed<float>(h, dtype, S, n, (float*)A, lda, sA, info, st, zero_upper, zero_info);
    if (dtype == MXF_F64) return potrf_typed<double>(h, dtype, S, n, (double*)A, lda, sA, info, st, zero_upper, zero_info, (double*)Linv_eager, ldie, eager_done,
                                                     (double*)Kacc, ldk, kcoef, kacc_done);
    MXF_FAIL(h, -2, "mxf_potrf: bad dtype %d", dtype);
}

int mxf_trsm_internal(mxf_ctx* h, int dtype, int transpose, int S, int64_t n, int64_t nrhs, const void* L, int64_t ldl,
                      int64_t sL, void* B, int64_t ldb, int64_t sB, int rhs_lower, hipStream_t st) {
    if (n <= 0 || nrhs <= 0 || S <= 0) return 0;
    if (dtype == MXF_F32) return trsm_typed<float>(h, dtype, transpose, S, n, nrhs, (const float*)L, ldl, sL, (float*)B, ldb, sB, rhs_lower, st);
    if (dtype == MXF_F64) return trsm_typed<double>(h, dtype, transpose, S, n, nrhs, (const double*)L, ldl, sL, (double*)B, ldb, sB, rhs_lower, st);
    MXF_FAIL(h, -2, "mxf_trsm: bad dtype %d", dtype);
}

namespace {

template <typename T>
__global__ void zero_block_kernel(T* __restrict__ P, int64_t rows, int64_t cols, int64_t ld, int64_t stride) {
    T* p = P + (int64_t)blockIdx.y * stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = (T)0;
}

// Log-depth blocked inverse of a lower-triangular matrix.  Level 0: all 64x64 diagonal blocks (one launch).  Level l merges pairs of
// bs-blocks:  inv([L11 0; L21 L22]) = [I11 0; -I22 L21 I11, I22]  with two batched MFMA GEMMs; the temporary (L21 I11)^T lives in the
// (unused, finally zeroed) upper-triangular mirror block of the output, so no extra workspace is needed.
template <typename T>
int trtri_typed(mxf_ctx* h, int dtype, int S, int64_t n, const T* L, int64_t ldl, int64_t sL, T* Li, int64_t ldi, int64_t sI, hipStream_t st) {
    if (n > 65535) MXF_FAIL(h, -3, "mxf_trtri: n too large");
    const int64_t nblk = (n + NB - 1) / NB;
    hipLaunchKernelGGL((trtri_diag_kernel<T>), dim3((unsigned)nblk, S), dim3(64), 0, st, L, ldl, sL, Li, ldi, sI, n);
    for (int64_t bs = NB; bs < n; bs *= 2) {
        const int64_t npairs_full = n / (2 * bs);                 // pairs whose second block is complete
        const int64_t rem0 = npairs_full * 2 * bs;                  // start of a possible ragged last pair
        for (int s = 0; s < S; ++s) {
            const T* Ls = L + (int64_t)s * sL;
            T* Is = Li + (int64_t)s * sI;
            if (npairs_full > 0) {
                const int64_t stL = 2 * bs * (ldl + 1), stI = 2 * bs * (ldi + 1);
                // tmpT (bs x bs, in the upper mirror block) = I11^T L21^T
                // (both products have a triangular left operand: from TRI_MIN-wide blocks on only the non-zero k range of each row tile is
                //  multiplied -- trtri(8192) 6.9 -> 5.0 ms; the small levels keep the plain split-K products, which fill the chip better)
                int rc = mxf_gemm_internal(h, dtype, 1, 1, bs, bs, bs, 1.0, Is, ldi, stI, Ls + bs * ldl, ldl, stL, 0.0, Is + bs, ldi, stI,
                                           (int)npairs_full, 0, st, 0, bs >= TRI_MIN ? 1 : 0);
                if (rc) return rc;
                // X21 = -I22 tmpT^T
                rc = mxf_gemm_internal(h, dtype, 0, 1, bs, bs, bs, -1.0, Is + bs * (ldi + 1), ldi, stI, Is + bs, ldi, stI, 0.0, Is + bs * ldi, ldi,
                                       stI, (int)npairs_full, 0, st, 0, bs >= TRI_MIN ? 2 : 0);
                if (rc) return rc;
                // the scratch blocks become part of the next level's I11 operand: they must be zero again
                hipLaunchKernelGGL((zero_block_kernel<T>), dim3((unsigned)((bs * bs + 255) / 256 > 1024 ? 1024 : (bs * bs + 255) / 256), (unsigned)npairs_full),
                                   dim3(256), 0, st, Is + bs, bs, bs, ldi, stI);
            }
            const int64_t b2 = n - rem0 - bs;                       // rows of the ragged second block of the last pair (if any)
            if (rem0 < n && b2 > 0) {
                const T* Lp = Ls + rem0 * (ldl + 1);
                T* Ip = Is + rem0 * (ldi + 1);
                int rc = mxf_gemm_internal(h, dtype, 1, 1, bs, b2, bs, 1.0, Ip, ldi, 0, Lp + bs * ldl, ldl, 0, 0.0, Ip + bs, ldi, 0, 1, 0, st, 0, bs >= TRI_MIN ? 1 : 0);
                if (rc) return rc;
                rc = mxf_gemm_internal(h, dtype, 0, 1, b2, bs, b2, -1.0, Ip + bs * (ldi + 1), ldi, 0, Ip + bs, ldi, 0, 0.0, Ip + bs * ldi, ldi, 0, 1, 0, st, 0, bs >= TRI_MIN ? 2 : 0);
                if (rc) return rc;
                hipLaunchKernelGGL((zero_block_kernel<T>), dim3((unsigned)((bs * b2 + 255) / 256 > 1024 ? 1024 : (bs * b2 + 255) / 256), 1), dim3(256), 0, st,
                                   Ip + bs, bs, b2, ldi, (int64_t)0);
            }
        }
    }
    if (n > 1) hipLaunchKernelGGL((zero_upper_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)n, S), dim3(256), 0, st, Li, n, ldi, sI);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

}  // namespace

int mxf_trtri_internal(mxf_ctx* h, int dtype, int S, int64_t n, const void* L, int64_t ldl, int64_t sL, void* Linv, int64_t ldi,
                       int64_t sI, hipStream_t st) {
    if (n <= 0 || S <= 0) return 0;
    if (dtype == MXF_F32) return trtri_typed<float>(h, dtype, S, n, (const float*)L, ldl, sL, (float*)Linv, ldi, sI, st);
    if (dtype == MXF_F64) return trtri_typed<double>(h, dtype, S, n, (const double*)L, ldl, sL, (double*)Linv, ldi, sI, st);
    MXF_FAIL(h, -2, "mxf_trtri: bad dtype %d", dtype);
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

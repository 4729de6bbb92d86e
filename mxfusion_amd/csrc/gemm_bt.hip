// f16x2 split product whose second operand is stored K-MAJOR:  C (M x N) = alpha * A (M x K) * Bt (K x N),  f32-equivalent accuracy
// (two scaled f16 terms per operand, three v_mfma_f32_32x32x16_f16 products: gemm_split.hip's format and arithmetic).
//
// Why (r06): the SVGP training call (svgp_regression.py:85-90) needs Kuf twice -- as the (m, k = n) operand of Psi2 = Kuf Kuf^T and as the
// (n, k = m) operand of T = H0 Kuf -- and until r05 wrote it twice, 8.6 GB of planes each; the second pass (1.8 ms, HBM-write bound) sat
// between the two products with the matrix pipe idle.  This kernel reads T's second operand from the FIRST set of planes: the planes of
// the (R = K rows, k' = N) operand "Bt", element (k, n) at ((n / 16) * R + k) * 16 + n % 16 -- 16 x 16 tiles [k][n], n minor.  The MFMA
// wants, per lane, eight CONSECUTIVE k of one column n; the tile has them 32 bytes apart.  gfx950's transposing LDS read does the turn:
//   * LDS-DMA (global_load_lds_dwordx4, lane-linear destination, free per-lane source) builds an image of 128-byte chunks [4 k][16 n];
//   * ds_read_b64_tr_b16: within a 16-lane group lane p ADDRESSES row p / 4, columns 4 (p % 4) .. + 3 of a chunk and RECEIVES column p of
//     its four rows -- two reads give the lane its eight k.  The chunks of one instruction's four lane groups are consecutive (512
//     contiguous bytes per instruction: conflict free).
// A (shared by the workgroup's eight waves) goes through LDS exactly as in gemm_split.hip's 256 x 256 kernel; so does Bt now, which ends
// that kernel's per-wave register ring of B fragments (48 registers) and its eight-fold redundant fragment fetch.  Four-slot ring of
// (A 16 KB + Bt 16 KB), requests three k blocks ahead as one stream ACROSS work items, one barrier per k block, every load an LDS-DMA
// request (4 per thread and k block).  What bounds it, and the alternatives that were measured: DESIGN.md section 4.2.
//
// U (optional, P = 1 of the SVGP call): the row  U[n] = uscale * sum_k w[k] Bt[k][n]  rides on the fragments that are in registers anyway:
// for every column strip ONE of its tm row-tile workgroups (rotating) also forms v_dot2_f32_f16 sums of its Bt fragments with the f16
// hi / lo planes of w (three products, like the MFMAs), its two row halves taking alternate k blocks.  (Until r05 U was a by-product of the
// second planes pass; every other pass over Kuf happens before w exists.)
#include "common.h"
#include "internal.h"
#include <string.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct BtArgs {
    const unsigned short* A; const unsigned short* Bt; float* C;
    int64_t M, N, K16;                   // K16 = number of 16-wide k blocks
    int64_t pA, pB, btR;                 // plane strides (elements); rows of the Bt operand (>= 16 K16)
    float alpha;
    int c_blk;                           // C in 16-column blocks: element (row, col) at ((col / 16) * M + row) * 16 + col % 16
    int64_t ldc;
    int64_t tm, tn, nwg;
    const float* ad0;                    // alpha *= ad0[0] (device scalar: the kernel variance of Gram planes)
    const unsigned* maxbits;             // alpha /= scale_from_maxbits(maxbits[0]) (the power-of-two scale of the A planes)
    const unsigned* maxbits2;            // the same for the Bt planes (nullptr: Gram planes, whose scale is known: alpha / ad0 carry it)
    unsigned* maxout;                    // atomicMax of the bit pattern of max |C| (nullptr: none)
    unsigned* sync; int sync_n;          // rendezvous of the tm row tiles of a column strip (gemm_split.hip: wg_rendezvous)
    int sync_every;                      // ... in every sync_every-th persistent round only: bounds the drift of the tiles that share a strip's lines in L2
    // U row: w as two f16 planes [2][16 K16] (hi, lo of w * scale_from_maxbits(uwmax[0])), U[n] = uscale * ad0[0] / that scale * sum
    const unsigned short* uw; const unsigned* uwmax; float* uout; float uscale;
};

__device__ __forceinline__ float bt_scale_from_maxbits(unsigned bits) {      // == gemm_split.hip's scale_from_maxbits
    const int ex = (int)((bits >> 23) & 0xff);
    if (ex == 0 || ex == 0xff) return 1.f;
    int e = 14 - (ex - 126);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

__device__ __forceinline__ int bt_lds_unit(int row, int kh) { return row * 2 + (kh ^ ((row >> 3) & 1)); }

constexpr unsigned long long BT_SYNC_LIMIT = 2000ull;       // wall_clock64 ticks (100 MHz): 20 us

__device__ __forceinline__ void bt_rendezvous(unsigned* ctr, unsigned n, int& patience) {      // see gemm_split.hip: a bounded pacing hint
    if (threadIdx.x == 0) {
        if (patience > 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
                if (wall_clock64() - t0 > BT_SYNC_LIMIT) { --patience; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == 2u * n)
                __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (__hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2u == 2u * n) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_barrier();
}

// one LDS-DMA request: 64 lanes x 16 bytes from per-lane global addresses to the 1 KB at LDS byte address `lds` (wave-uniform, in M0)
// (wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset: two address registers per thread for the whole kernel)
__device__ __forceinline__ void bt_dma16(const void* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

// acc + sum of the eight f16 products of two 16-byte fragments (four v_dot2_f32_f16)
__device__ __forceinline__ float bt_dot8(u32x4 a, u32x4 b, float acc) {
    const f16x8 av = __builtin_bit_cast(f16x8, a), bv = __builtin_bit_cast(f16x8, b);
    acc = __builtin_amdgcn_fdot2(f16x2{av[0], av[1]}, f16x2{bv[0], bv[1]}, acc, false);
    acc = __builtin_amdgcn_fdot2(f16x2{av[2], av[3]}, f16x2{bv[2], bv[3]}, acc, false);
    acc = __builtin_amdgcn_fdot2(f16x2{av[4], av[5]}, f16x2{bv[4], bv[5]}, acc, false);
    acc = __builtin_amdgcn_fdot2(f16x2{av[6], av[7]}, f16x2{bv[6], bv[7]}, acc, false);
    return acc;
}

constexpr int BT_KMAX16 = 128;           // U row: w planes of up to 2048 k live in LDS

// 256 x 256 tile, 512 threads: wave = 4 h + w owns rows [128 h, + 128) x columns [64 w, + 64) = 4 x 2 MFMA tiles (128 accumulators).
template <bool WU, bool RA = false, bool PM = false, int DG = 0>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_bt_kernel(BtArgs g) {
    constexpr int NU = 512;                          // 16-byte units of one plane's (256 x 16) slab
    __shared__ u32x4 sA[4][2][NU];                   // [ring slot][plane][unit]: A rows, gemm_split.hip's swizzled image
    __shared__ u32x4 sB[4][2][NU];                   // Bt: chunk c = ((wq * 2 + y) * 2 + h) * 4 + grp of 128 bytes [4 k][16 n]
    __shared__ u32x4 sW[WU ? 2 * BT_KMAX16 * 2 : 1]; // w planes [plane][16 K16 halves]
    __shared__ float sU[WU ? 512 : 1];               // the two row halves' shares of U for the item's 256 columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wq = wave & 3, wh = wave >> 2;
    const int li = lane & 31, lk = lane >> 5;
    int patience = 2;
    float cmax = 0.f;
    if constexpr (WU) {
        if (g.uout) {
            const int nunits = (int)(g.K16 * 2);          // 16-byte units per plane
            for (int i = tid; i < 2 * nunits; i += 512) {
                const int p = i / nunits, u = i % nunits;
                sW[p * (BT_KMAX16 * 2) + u] = *reinterpret_cast<const u32x4*>(g.uw + (int64_t)p * g.K16 * 16 + u * 8);
            }
        }
        __syncthreads();
    }
    float alpha = g.alpha;
    if (g.ad0) alpha *= g.ad0[0];
    if (g.maxbits) alpha /= bt_scale_from_maxbits(g.maxbits[0]);
    if (g.maxbits2) alpha /= bt_scale_from_maxbits(g.maxbits2[0]);

    // A: thread t fills unit t of each plane's slab (row t >> 1; the XOR swizzle of the two k halves is applied to the source)
    const int drow = tid >> 1, dkh = (tid & 1) ^ ((drow >> 3) & 1);
    // Bt: thread t fills unit t = 8 c + s of the image: chunk c = (wq', y, h, grp), piece s = (row j = s >> 1 of the chunk, n half s & 1)
    const int bc = tid >> 3, bs = tid & 7;
    const int b_nb = 4 * (bc >> 4) + 2 * ((bc >> 3) & 1) + (bc & 1);              // n16 block inside the strip: 4 wq' + 2 y + (grp & 1)
    const int b_kl = 8 * ((bc >> 1) & 1) + 4 * ((bc >> 2) & 1) + (bs >> 1);       // k inside the block: 8 (grp >> 1) + 4 h + j
    const unsigned voff_b = (unsigned)((((int64_t)b_nb * g.btR + b_kl) * 16 + 8 * (bs & 1)) * 2);      // bytes (host checks 16 btR * 32 < 2^31)
    const unsigned voff_a = (unsigned)((drow * 16 + dkh * 8) * 2);
    // fragment reads: A units; Bt byte offset of this lane inside a plane image (instruction (y, h) adds (2 y + h) * 512)
    // (row = 128 wh + 32 x + li: the swizzle bit (row >> 3) & 1 does not depend on x, so fragment x sits 64 units behind fragment 0 -- one
    //  address register and immediate offsets)
    const int ua0_ = bt_lds_unit(128 * wh + li, lk);
#define ua_(x) (ua0_ + 64 * (x))
    const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sA[0][0][wave * 64]);
    const unsigned ldsB = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)&sB[0][0][wave * 64]);
    const unsigned toff_blk = (unsigned)((((int64_t)(4 * wq) * g.M + 128 * wh + li) * 16 + 4 * lk) * 4);      // bytes (host: 16 M * 64 < 2^31)
    const int rb_h = (wq * 2048 + (lane >> 4) * 128 + ((lane & 15) >> 2) * 32 + (lane & 3) * 8) / 2;      // in halves

    // One STREAM of k blocks over all of this workgroup's items (blockIdx.x, + gridDim.x, ...): the requests run three blocks ahead of the
    // multiplies ACROSS item boundaries, so an item's first blocks arrive under the previous item's last multiplies and its epilogue -- a
    // per-item pipeline fill (two memory latencies with nothing in flight, ~3 % of a K = 1024 item) never happens.  Ring slot = stream
    // position modulo 4 (the fragments are read from LDS, so nothing here needs a compile-time ring index).
    // (32-bit index arithmetic throughout: the 64-bit divisions of a first form expanded to branchy software routines in the per-item paths,
    //  with spill reloads whose compiler-inserted s_waitcnt vmcnt(0) drained the request stream once per item)
    const int nk = (int)g.K16;
    const unsigned nwg = (unsigned)g.nwg, tmu = (unsigned)g.tm;
    const int nit = (int)((nwg - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int nsteps = nit * nk;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7;
    auto item_of = [&](int i) -> unsigned {         // XCD-aware mapping: every XCD owns a contiguous run of items (gemm_split.hip)
        const unsigned wid = blockIdx.x + (unsigned)i * gridDim.x;
        const unsigned xcd = wid & 7, jx = wid >> 3;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + jx;
    };
    // request side
    int q_it = 0, q_kb = 0;
    unsigned q_slot = 0;
    const unsigned short* qa = nullptr;
    const unsigned short* qb = nullptr;
    auto q_enter = [&]() {
        const unsigned wid = item_of(q_it);
        const unsigned tn_ = wid / tmu, tm_ = wid - tn_ * tmu;      // the row tiles of one column strip are neighbours
        qa = g.A + (int64_t)tm_ * 256 * 16;           // wave-uniform bases: SGPRs
        qb = g.Bt + ((int64_t)tn_ * 16) * g.btR * 16;
    };
    // (inline asm, not __builtin_amdgcn_global_load_lds: with the builtin the compiler tracks the LDS-DMA writes itself and puts an
    //  s_waitcnt vmcnt(0) in front of the fragment reads at the loop header -- it cannot see the counted waits below --, i.e. one full
    //  memory latency per trip with nothing in flight)
    auto q_issue = [&]() {
        bt_dma16(qa + (int64_t)q_kb * g.M * 16, voff_a, ldsA + q_slot * 16384);
        bt_dma16(qa + g.pA + (int64_t)q_kb * g.M * 16, voff_a, ldsA + q_slot * 16384 + 8192);
        bt_dma16(qb + (int64_t)q_kb * 256, voff_b, ldsB + q_slot * 16384);
        bt_dma16(qb + g.pB + (int64_t)q_kb * 256, voff_b, ldsB + q_slot * 16384 + 8192);
        q_slot = (q_slot + 1) & 3;
        if (++q_kb == nk) { q_kb = 0; ++q_it; if (q_it < nit) q_enter(); }
    };
    // all but the block requested last (4 requests) have landed -- this wave's share; the barrier extends it to the workgroup.  (vmcnt
    // retires in order and counts the epilogue's stores too: a wait behind an epilogue also drains that item's stores -- conservative.)
#define BT_WAIT(NSTR) do { asm volatile("s_waitcnt vmcnt(" NSTR ")" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define HF(v) __builtin_bit_cast(f16x8, v)
#define BT_TR(ptr) __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr)))
    f32x16 c[4][2];
    float ua0 = 0.f, ua1 = 0.f;
    int c_kb = 0, c_it = 0;
    int64_t m0 = 0, n0 = 0;
    unsigned c_slot = 0;
    bool uitem = false;
    // FOUR ring slots.  Default (RA = false): the requests run THREE blocks ahead, a step reads its own fragments (Bt + A fragment 0 at the
    // top, the other A fragments two multiplies ahead of their use).  RA = true (probe builds): requests two blocks ahead and the NEXT
    // block's Bt fragments + first A fragment are read into a second register set under this block's MFMAs -- measured 0.15 ms slower per
    // 32-sample step.  (r06 PMC of the first form -- three slots, all sixteen fragments read behind the barrier: matrix pipe 0.54 busy at
    // 1.80 GHz, i.e. stalled, not power-bound; the r05 kernel, whose B fragments sat in a register ring: 0.69 at 1.56; this form: 0.64 at 1.62.)
    u32x4 fb0[2][2], fa0[2], fb1[2][2], fa1[2];
    u32x4 dga[6];      // (the A fragments 1-3 of the current block; a named array only so that the DG diagnostics can keep them)
#define BT_READ_B(FB, SLOTV)                                                                                                        \
    do {                                                                                                                            \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                             \
            const unsigned short* sb_ = reinterpret_cast<const unsigned short*>(&sB[SLOTV][p][0]) + rb_h;                           \
            _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                         \
                const u32x2 v0_ = BT_TR(sb_ + (2 * y) * 256), v1_ = BT_TR(sb_ + (2 * y + 1) * 256);                                 \
                FB[y][p] = u32x4{v0_[0], v0_[1], v1_[0], v1_[1]};                                                                   \
            }                                                                                                                       \
        }                                                                                                                           \
    } while (0)
#define BT_READ_A0(FA, SLOTV) do { FA[0] = sA[SLOTV][0][ua_(0)]; FA[1] = sA[SLOTV][1][ua_(0)]; } while (0)
#define BT_MM(x, AH, AL, FB)                                                                                                        \
    do {                                                                                                                            \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                             \
            c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AL), c[x][y], 0, 0, 0);          /* hi' lo */         \
            c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][1]), HF(AH), c[x][y], 0, 0, 0);          /* lo' hi */         \
            c[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AH), c[x][y], 0, 0, 0);          /* hi' hi */         \
        }                                                                                                                           \
    } while (0)
#define BT_MM2(x0, AH0, AL0, x1, AH1, AL1, FB)                                                                                      \
    do {                                                                                                                            \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                             \
            c[x0][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AL0), c[x0][y], 0, 0, 0);                            \
            c[x1][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AL1), c[x1][y], 0, 0, 0);                            \
        }                                                                                                                           \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                             \
            c[x0][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][1]), HF(AH0), c[x0][y], 0, 0, 0);                            \
            c[x1][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][1]), HF(AH1), c[x1][y], 0, 0, 0);                            \
        }                                                                                                                           \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                                                             \
            c[x0][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AH0), c[x0][y], 0, 0, 0);                            \
            c[x1][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(FB[y][0]), HF(AH1), c[x1][y], 0, 0, 0);                            \
        }                                                                                                                           \
    } while (0)
    // one stream position: CUR holds block s_ (its Bt fragments and A fragment 0), NXT receives block s_ + 1
#define BT_STEP(FB, FA, NFB, NFA)                                                                                                   \
    do {                                                                                                                            \
        if (c_kb == 0) {                             /* a new item (workgroup-uniform) */                                           \
            const unsigned wid = item_of(c_it);                                                                                     \
            const unsigned tile_n = wid / tmu, tile_m = wid - tile_n * tmu;                                                         \
            m0 = (int64_t)tile_m * 256; n0 = (int64_t)tile_n * 256;                                                                 \
            uitem = WU && g.uout != nullptr && ((tile_n + (tile_n >> 3) + (tile_n >> 6)) % tmu) == tile_m;                          \
            if (g.sync && (c_it % g.sync_every) == 0) bt_rendezvous(g.sync + wid / g.sync_n, (unsigned)g.sync_n, patience);         \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                                           \
                _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                       \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) c[x][y][r] = 0.f;                                                \
            ua0 = 0.f; ua1 = 0.f;                                                                                                   \
        }                                                                                                                           \
        const bool more = s_ + 3 < nsteps;                                                                                          \
        if (more) q_issue();                         /* block s_ + 3 into the slot block s_ - 1 left */                             \
        asm volatile("" ::: "memory");               /* the fragment reads stay behind the requests */                              \
        const unsigned nslot = (c_slot + 1) & 3;                                                                                    \
        if constexpr (!RA) {                                                                                                        \
            if (DG != 2 || c_kb == 0) BT_READ_B(FB, c_slot);      /* DG (probe builds): timing diagnostics with WRONG results -- 1: the A fragments, */ \
            if (DG != 1 || c_kb == 0) BT_READ_A0(FA, c_slot);     /* 2: the Bt fragments are read for an item's first k block only                   */ \
        }                                                                                                                           \
        if constexpr (WU) {                                                                                                         \
            if (uitem && (((int)c_kb & 1) == wh)) {      /* wave-uniform; FIRST: few registers are live here */                        \
                const u32x4 wh_ = sW[2 * (int)c_kb + lk], wl_ = sW[BT_KMAX16 * 2 + 2 * (int)c_kb + lk];                             \
                ua0 = bt_dot8(FB[0][1], wh_, ua0); ua0 = bt_dot8(FB[0][0], wl_, ua0); ua0 = bt_dot8(FB[0][0], wh_, ua0);            \
                ua1 = bt_dot8(FB[1][1], wh_, ua1); ua1 = bt_dot8(FB[1][0], wl_, ua1); ua1 = bt_dot8(FB[1][0], wh_, ua1);            \
            }                                                                                                                       \
        }                                                                                                                           \
        if constexpr (PM) {                                                                                                         \
            /* (measured alternative, probe knob MXF_BT_PM) product-major in PAIRS of row fragments: four independent MFMAs between two that share an */ \
            /* accumulator -- same box, 32-sample step 22.38-22.46 ms against 22.31-22.35: back-to-back accumulation is not what stalls the pipe    */ \
            u32x4 a1h = sA[c_slot][0][ua_(1)], a1l = sA[c_slot][1][ua_(1)];                                                           \
            u32x4 a2h = sA[c_slot][0][ua_(2)], a2l = sA[c_slot][1][ua_(2)];                                                           \
            u32x4 a3h = sA[c_slot][0][ua_(3)], a3l = sA[c_slot][1][ua_(3)];                                                           \
            BT_MM2(0, FA[0], FA[1], 1, a1h, a1l, FB);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            if constexpr (RA) { if (s_ + 1 < nsteps) { BT_READ_B(NFB, nslot); BT_READ_A0(NFA, nslot); } }                           \
            BT_MM2(2, a2h, a2l, 3, a3h, a3l, FB);                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        } else {                                                                                                                    \
        /* the A fragments are requested two multiplies ahead of their use                                                           */ \
        if (DG != 1 || c_kb == 0) { dga[0] = sA[c_slot][0][ua_(1)]; dga[1] = sA[c_slot][1][ua_(1)]; dga[2] = sA[c_slot][0][ua_(2)]; dga[3] = sA[c_slot][1][ua_(2)]; } \
        u32x4 a1h = dga[0], a1l = dga[1];                                                                                           \
        u32x4 a2h = dga[2], a2l = dga[3];                                                                                           \
        BT_MM(0, FA[0], FA[1], FB);                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);           /* (pins the order: the scheduler otherwise hoists every read to the top) */    \
        if (DG != 1 || c_kb == 0) { dga[4] = sA[c_slot][0][ua_(3)]; dga[5] = sA[c_slot][1][ua_(3)]; }                               \
        u32x4 a3h = dga[4], a3l = dga[5];                                                                                           \
        BT_MM(1, a1h, a1l, FB);                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        if constexpr (RA) { if (s_ + 1 < nsteps) { BT_READ_B(NFB, nslot); BT_READ_A0(NFA, nslot); } }     /* block s_ + 1 landed at the last barrier */    \
        BT_MM(2, a2h, a2l, FB);                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        BT_MM(3, a3h, a3l, FB);                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        }                                                                                                                           \
        if constexpr (RA) { if (more) BT_WAIT("4"); else BT_WAIT("0"); }   /* block s_ + 2 has landed (workgroup-uniform branch) */  \
        else { if (more) BT_WAIT("8"); else if (s_ + 2 < nsteps) BT_WAIT("4"); else BT_WAIT("0"); }      /* block s_ + 1 */            \
        c_slot = nslot;                                                                                                             \
        if (++c_kb == nk) { c_kb = 0; ++c_it; finish_item(); }                                                                      \
        ++s_;                                                                                                                       \
    } while (0)
    auto finish_item = [&]() {
        if constexpr (WU) {
            if (uitem) {          // workgroup-uniform
                ua0 += __shfl_xor(ua0, 32, 64);              // the two k halves of the fragment
                ua1 += __shfl_xor(ua1, 32, 64);
                if (lane < 32) { sU[256 * wh + 64 * wq + li] = ua0; sU[256 * wh + 64 * wq + 32 + li] = ua1; }
                __syncthreads();
                if (tid < 256) {
                    float sc = g.uscale * (g.ad0 ? g.ad0[0] : 1.f) / bt_scale_from_maxbits(g.uwmax[0]);
                    if (g.maxbits2) sc /= bt_scale_from_maxbits(g.maxbits2[0]);
                    g.uout[n0 + tid] = (sU[tid] + sU[256 + tid]) * sc;
                }
            }
        }
        // D = Bt^T A^T: accumulator register r of tile (x, y) is C[m0 + 128 wh + 32 x + li][n0 + 64 wq + 32 y + 8 (r >> 2) + 4 lk + (r & 3)]
        if (g.c_blk) {
            // blocked layout, element (row, col) at ((col / 16) * M + row) * 16 + col % 16: one 32-bit per-thread byte offset (toff_blk) + a
            // wave-uniform base per (x, y, q) -- the 32 store addresses cost one register, not 64
            char* const cb = reinterpret_cast<char*>(g.C) + ((n0 >> 4) * g.M + m0) * 64;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        char* const sb = cb + ((int64_t)(2 * y + (q >> 1)) * g.M + 32 * x) * 64 + 32 * (q & 1);      // wave-uniform
                        const f32x4 v = {alpha * c[x][y][4 * q], alpha * c[x][y][4 * q + 1], alpha * c[x][y][4 * q + 2], alpha * c[x][y][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(sb + toff_blk) = v;
                        if (g.maxout) cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    }
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int64_t row = m0 + 128 * wh + 32 * x + li;
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t col = n0 + 64 * wq + 32 * y + 8 * q + 4 * lk;
                        float* p = g.C + row * g.ldc + col;
                        const f32x4 v = {alpha * c[x][y][4 * q], alpha * c[x][y][4 * q + 1], alpha * c[x][y][4 * q + 2], alpha * c[x][y][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(p) = v;
                        if (g.maxout) cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    }
            }
        }
    };
    if (nsteps > 0) {
        q_enter();
        q_issue();
        q_issue();
        q_issue();                                   // (K >= 48: every item has at least three blocks)
        BT_WAIT("8");                                // block 0 landed
        if constexpr (RA) {
            BT_READ_B(fb0, 0); BT_READ_A0(fa0, 0);
            BT_WAIT("4");                            // block 1 landed
        }
    }
    int s_ = 0;
    if constexpr (RA) {
        while (s_ + 1 < nsteps) {                    // two positions per trip: the fragment sets alternate at compile time
            BT_STEP(fb0, fa0, fb1, fa1);
            BT_STEP(fb1, fa1, fb0, fa0);
        }
        if (s_ < nsteps) BT_STEP(fb0, fa0, fb1, fa1);
    } else {
        while (s_ < nsteps) BT_STEP(fb0, fa0, fb1, fa1);
    }
#undef ua_
#undef BT_STEP
#undef BT_MM2
#undef BT_MM
#undef BT_READ_A0
#undef BT_READ_B
#undef BT_TR
#undef HF
#undef BT_WAIT
    if (g.maxout) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && __builtin_bit_cast(unsigned, cmax) > *(volatile unsigned*)g.maxout) atomicMax(g.maxout, __builtin_bit_cast(unsigned, cmax));
    }
}

// w (K floats) -> two f16 planes of w * scale (scale from max |w|, which is also written to maxword); one workgroup
__global__ __launch_bounds__(256) void bt_wsplit_kernel(int64_t K, int64_t Kp, const float* __restrict__ w, unsigned short* __restrict__ planes,
                                                        unsigned* __restrict__ maxword) {
    __shared__ unsigned wm[4];
    unsigned m = 0;
    for (int64_t i = threadIdx.x; i < K; i += 256) { const unsigned b = __builtin_bit_cast(unsigned, w[i]) & 0x7fffffffu; m = b > m ? b : m; }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    for (int i = 0; i < 4; ++i) m = wm[i] > m ? wm[i] : m;
    if (threadIdx.x == 0) maxword[0] = m;
    const float sc = bt_scale_from_maxbits(m);
    for (int64_t i = threadIdx.x; i < Kp; i += 256) {
        const float x = i < K ? w[i] * sc : 0.f;
        const _Float16 fh = (_Float16)x;
        const _Float16 fl = (_Float16)(x - (float)fh);
        planes[i] = __builtin_bit_cast(unsigned short, fh);
        planes[Kp + i] = __builtin_bit_cast(unsigned short, fl);
    }
}

}  // namespace

bool mxf_gemm_bt_ok(int64_t M, int64_t N, int64_t K) { return M > 0 && N > 0 && K >= 48 && (M % 256) == 0 && (N % 256) == 0 && (K % 16) == 0; }

// C (M x N) = alpha * ad0[0] / scale(maxbits) * A (M x K) * Bt (K x N) from f16x2 planes: A planes as in gemm_split.hip ((m, k): ((k / 16) * M
// + m) * 16 + k % 16, plane stride pA), Bt = the planes of the (btR >= K rows, k' = N) operand ((n / 16) * btR + k) * 16 + n % 16, plane
// stride pB.  c_blocked: C in 16-column blocks (ldc == N).  w (K floats, device) + U (N floats): U[n] = uscale * ad0[0] * sum_k w[k] Bt[k][n]
// (Bt in the planes' units); wscratch: 2 K halves + one word, caller-owned.  reserve_cus: workgroups = 256 - reserve_cus (one per CU).
int mxf_gemm_bt_internal(mxf_ctx* h, int64_t M, int64_t N, int64_t K, double alpha, const unsigned short* A, int64_t pA, const unsigned short* Bt,
                         int64_t pB, int64_t btR, float* C, int64_t ldc, int c_blocked, hipStream_t st, int reserve_cus, const float* ad0,
                         const unsigned* maxbits, unsigned* maxout, const float* w, float* U, double uscale, void* wscratch, const unsigned* maxbits2) {
    if (c_blocked && M * 16 * 64 >= (1ll << 31)) MXF_FAIL(h, -3, "mxf_gemm_bt: too many rows for the blocked output");
    if (btR * 16 * 32 >= (1ll << 31)) MXF_FAIL(h, -3, "mxf_gemm_bt: the K-major operand has too many rows");
    if (!mxf_gemm_bt_ok(M, N, K) || btR < K) MXF_FAIL(h, -2, "mxf_gemm_bt: needs M %% 256 == 0, N %% 256 == 0, K %% 16 == 0, K >= 48");
    if (c_blocked && ldc != N) MXF_FAIL(h, -2, "mxf_gemm_bt: the blocked output layout needs ldc == N");
    if ((ldc % 4) != 0 || (((uintptr_t)C) % 16) != 0) MXF_FAIL(h, -2, "mxf_gemm_bt: C must be 16-byte aligned with ldc %% 4 == 0");
    if (U && (!w || !wscratch || K / 16 > BT_KMAX16)) MXF_FAIL(h, -2, "mxf_gemm_bt: the U row needs w, scratch and K <= %d", BT_KMAX16 * 16);
    BtArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.Bt = Bt; g.C = C; g.M = M; g.N = N; g.K16 = K / 16; g.pA = pA; g.pB = pB; g.btR = btR;
    g.alpha = (float)alpha; g.c_blk = c_blocked; g.ldc = ldc; g.tm = M / 256; g.tn = N / 256; g.nwg = g.tm * g.tn;
    g.ad0 = ad0; g.maxbits = maxbits; g.maxbits2 = maxbits2; g.maxout = maxout;
    if (U) {
        unsigned short* wp = (unsigned short*)wscratch;
        unsigned* wmax = (unsigned*)(wp + 2 * K);
        hipLaunchKernelGGL(bt_wsplit_kernel, dim3(1), dim3(256), 0, st, K, K, w, wp, wmax);
        g.uw = wp; g.uwmax = wmax; g.uout = U; g.uscale = (float)uscale;
    }
    int64_t grid = (int64_t)(256 - (reserve_cus > 0 ? reserve_cus : 0)) / 8 * 8;
    if (grid < 8) grid = 8;
    if (g.nwg <= grid) grid = g.nwg;
    // The tm row tiles of a column strip (tm consecutive items of one XCD's run, taken in the same persistent round by tm different
    // workgroups) share the strip's Bt lines in their XCD's L2 -- as long as they run in step.  Nothing keeps them there: with no pacing
    // at all they drift apart and every one fetches the strip from HBM itself (r06 PMC: 50 GB of traffic per launch against 26 GB of
    // operands + output).  A bounded rendezvous (gemm_split.hip) at the start of an item re-aligns them.  Same box, T shape, traffic per
    // launch | 32-sample step: never 49.9 GB | 22.04-22.09 ms; every item 23.1 GB | 22.31; every 4th round 32.1 GB | 22.07; 8th 33.0 | 22.02-22.06;
    // 16th 34.1 | 22.06 (tests/probes/r06_bt_sync.sh).  Every fourth round: two thirds of the avoidable traffic gone at no cost in time.
    // (MXF_BT_SYNC = rounds between two rendezvous, 0 = never; probe builds.)
    static const int sync_env = (int)MXF_KNOB("MXF_BT_SYNC", 4);
    const int64_t q = g.nwg / 8, per_xcd = grid / 8;
    if (sync_env && g.tm >= 2 && g.nwg % 8 == 0 && g.nwg >= 16 && q % g.tm == 0 && per_xcd % g.tm == 0 && (g.nwg <= grid || g.nwg % grid == 0)) {
        g.sync = mxf_gsync(h, (unsigned)(g.nwg / g.tm));
        g.sync_n = (int)g.tm;
        g.sync_every = sync_env > 0 ? sync_env : 1;
    }
#ifdef MXF_PROBES
    // RA (register read-ahead of the next block's Bt fragments + first A fragment, requests two blocks ahead) against the default (fragments
    // read at the top of their own step, requests THREE blocks ahead): same box, 32-sample step 22.69-22.80 ms with it, 22.57 without
    static const int dg_env = (int)MXF_KNOB("MXF_BT_DIAG", 0);
    if (dg_env == 1 || dg_env == 2) {      // timing diagnostics, WRONG results
        if (dg_env == 1) hipLaunchKernelGGL((gemm_f16x2_bt_kernel<true, false, false, 1>), dim3((unsigned)grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_f16x2_bt_kernel<true, false, false, 2>), dim3((unsigned)grid), dim3(512), 0, st, g);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    static const int pm_env = (int)MXF_KNOB("MXF_BT_PM", 0);
    static const int ra_env = (int)MXF_KNOB("MXF_BT_RA", 0);
    if (pm_env) {
        if (U) hipLaunchKernelGGL((gemm_f16x2_bt_kernel<true, false, true>), dim3((unsigned)grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_f16x2_bt_kernel<false, false, true>), dim3((unsigned)grid), dim3(512), 0, st, g);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    if (ra_env) {
        if (U) hipLaunchKernelGGL((gemm_f16x2_bt_kernel<true, true>), dim3((unsigned)grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_f16x2_bt_kernel<false, true>), dim3((unsigned)grid), dim3(512), 0, st, g);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
#endif
    if (U) hipLaunchKernelGGL((gemm_f16x2_bt_kernel<true>), dim3((unsigned)grid), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_f16x2_bt_kernel<false>), dim3((unsigned)grid), dim3(512), 0, st, g);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// C ABI: C (M x N) = alpha * A (M x K) * Bt (K x N) from split operands -- A_planes = mxf_f16x2_split of A (M x K), Bt_planes = mxf_f16x2_split
// of the (K x N) matrix Bt ITSELF (its rows are the contraction index: no transposed copy of it is ever made).  blocked: C in 16-column
// blocks (mxf_gemm_f16x2_planes' lower_only = 2 layout).  w (K floats) and U (N floats), both or neither: U[n] = sum_k w[k] Bt[k][n].
extern "C" int mxf_gemm_f16x2_planes_kmajor(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                                            const void* Bt_planes, const void* Bt_maxword, void* C, int blocked, const void* w, void* U, void* stream) {
    if (!h) return -1;
    if (!A_planes || !Bt_planes || !A_maxword || !Bt_maxword || !C || ((w == nullptr) != (U == nullptr))) MXF_FAIL(h, -2, "mxf_gemm_f16x2_planes_kmajor: bad argument");
    if (!mxf_gemm_bt_ok(M, N, K)) MXF_FAIL(h, -3, "mxf_gemm_f16x2_planes_kmajor: needs M %% 256 == 0, N %% 256 == 0, K %% 16 == 0, K >= 48");
    void* ws = nullptr;
    if (U) {
        ws = mxf_ws(h, mxf_align((2 * (size_t)K + 8) * 2));
        if (!ws) MXF_FAIL(h, -4, "mxf_gemm_f16x2_planes_kmajor: cannot allocate scratch");
    }
    return mxf_gemm_bt_internal(h, M, N, K, alpha, (const unsigned short*)A_planes, (int64_t)mxf_split_plane_elems(M, K), (const unsigned short*)Bt_planes,
                                (int64_t)mxf_split_plane_elems(K, N), K, (float*)C, N, blocked ? 1 : 0, (hipStream_t)stream, 0, nullptr,
                                (const unsigned*)A_maxword, nullptr, (const float*)w, (float*)U, 1.0, ws, (const unsigned*)Bt_maxword);
}

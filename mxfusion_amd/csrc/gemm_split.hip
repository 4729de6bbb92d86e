// f32-accurate GEMM on the bf16 matrix pipe of gfx950:  C = alpha * A * B^T (+ beta * C),  A (M x K), B (N x K), f32 in / f32 out.
//
// The f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the VECTOR rate, 1/16 of v_mfma_f32_32x32x16_bf16.  Every f32 operand is
// therefore split EXACTLY into three bf16 terms  x = h + m + l  (8 + 8 + 8 significand bits, round-to-nearest at each step:
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)), and the product is formed from the six bf16 x bf16 products whose weight
// is >= 2^-16 of the leading one,
//        x y  ~=  m m' + h l' + l h' + h m' + m h' + h h'        (dropped: m l', l m', l l'  <= 2^-23 |x y|)
// accumulated in f32 inside the MFMA, i.e. the same product accuracy and the same accumulator as the f32 MFMA, at 6/16 of its
// cost (peak 2.5 PF / 6 = 417 TF vs 157 TF).  This is an f32 GEMM (Ozaki-style splitting), not a reduced-precision one: the
// parity tests bound its error against float64 exactly as for the plain f32 kernel.
//
// Operand storage ("planes"): plane p of an (R x K) operand holds the p-th bf16 term in k16-blocked order
//        element (r, k)  ->  ((k / 16) * R + r) * 16 + k % 16
// so that the (128 rows x 16 k) slab a workgroup needs per MFMA step is ONE contiguous 4 KB run: the loader is a linear 16-byte
// copy per thread and the LDS image equals the global one (with an XOR swizzle of the two 16-byte halves of a row, which makes
// the ds_read_b128 fragment reads conflict free).  A-fragment of v_mfma_f32_32x32x16_bf16: lane l -> A[i = l & 31][k = 8 (l >> 5) .. +7]
// = one 16-byte unit; B likewise (both operands are k-contiguous: "NT" form only -- the SVGP step has both Kuf and Kfu).
//
// Second operand format ("f16x2", NP = 2): x * s = hi + lo with two f16 terms (11 + 11 significand bits), s a power of two that puts the
// operand's largest magnitude at [2^13, 2^14] -- near the top of the f16 range, so that lo keeps its 11 bits down to 2^-18 of the
// maximum and its absolute error never exceeds 2^-39 of it.  Three products (hi hi' + hi lo' + lo hi'; dropped lo lo' <= 2^-22 |x y|,
// representation error <= 2^-23 |x|): the product accuracy of the f32 MFMA at 3/16 of its cost instead of 6/16 (peak 2.5 PF / 3).
// The scale of a Gram operand is known in closed form (unit-variance covariances are <= 1: planes hold k / variance * 2^14); the scale
// of a general operand comes from its max-abs word (mxf_maxabs_internal), read by the splitter and by the GEMM epilogue.
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBM = 128, SBN = 128, SNT = 256;   // one 16-wide k block per MFMA step

// ------------------------------------------------------------------------------------------------ f32 -> three bf16 planes
// X (R x K, row stride ld) -> planes; K is padded with zeros to a multiple of 16 (Kp).  One block: 64 rows x 64 k.
// power-of-two scale that puts |x| <= max at [2^13, 2^14]; max given as the bit pattern of a non-negative float
__device__ __forceinline__ float scale_from_maxbits(unsigned bits) {
    const int ex = (int)((bits >> 23) & 0xff);                 // biased exponent of the maximum: max in [2^(ex-127), 2^(ex-126))
    if (ex == 0 || ex == 0xff) return 1.f;                      // zero / denormal / non-finite maximum: leave unscaled
    int e = 14 - (ex - 126);                                    // max * 2^e in [2^13, 2^14)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(int64_t R, int64_t K, const float* __restrict__ X, int64_t ld,
                                                           unsigned short* __restrict__ P, int64_t pstride, const unsigned* __restrict__ maxbits) {
    __shared__ float tile[64][68];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64, k0 = (int64_t)blockIdx.x * 64;
    // coalesced read: 16 threads x float4 per row
    for (int it = 0; it < 4; ++it) {
        const int row = it * 16 + (tid >> 4), c = (tid & 15) * 4;
        const int64_t r = r0 + row, k = k0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R) {
            if (k + 3 < K && ((ld & 3) == 0) && (((uintptr_t)X & 15) == 0)) v = *reinterpret_cast<const float4*>(X + r * ld + k);
            else {
                if (k + 0 < K) v.x = X[r * ld + k + 0];
                if (k + 1 < K) v.y = X[r * ld + k + 1];
                if (k + 2 < K) v.z = X[r * ld + k + 2];
                if (k + 3 < K) v.w = X[r * ld + k + 3];
            }
        }
        tile[row][c + 0] = v.x; tile[row][c + 1] = v.y; tile[row][c + 2] = v.z; tile[row][c + 3] = v.w;
    }
    __syncthreads();
    const int row = tid & 63, kbl = tid >> 6;            // one (row, 16-k block) unit per thread
    const int64_t r = r0 + row, kb = k0 / 16 + kbl;
    const int64_t Kp16 = (K + 15) / 16;
    if (r >= R || kb >= Kp16) return;
    unsigned short h[16], m[16], l[16];
    const float sc = (NP == 2 && maxbits) ? scale_from_maxbits(maxbits[0]) : 1.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float x = tile[row][kbl * 16 + j];
        if (NP == 2) {
            const float xs = x * sc;
            const _Float16 fh = (_Float16)xs;
            const _Float16 fl = (_Float16)(xs - (float)fh);
            h[j] = __builtin_bit_cast(unsigned short, fh); m[j] = __builtin_bit_cast(unsigned short, fl); l[j] = 0;
            continue;
        }
        const __bf16 bh = (__bf16)x;
        const float r1 = x - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const float r2 = r1 - (float)bm;
        const __bf16 bl = (__bf16)r2;
        h[j] = __builtin_bit_cast(unsigned short, bh); m[j] = __builtin_bit_cast(unsigned short, bm); l[j] = __builtin_bit_cast(unsigned short, bl);
    }
    const int64_t off = (kb * R + r) * 16;
    auto put = [&](unsigned short* dst, const unsigned short (&v)[16]) {
        u32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16); b[j] = (unsigned)v[8 + 2 * j] | ((unsigned)v[8 + 2 * j + 1] << 16); }
        *reinterpret_cast<u32x4*>(dst) = a;
        *reinterpret_cast<u32x4*>(dst + 8) = b;
    };
    put(P + off, h);
    put(P + pstride + off, m);
    if (NP == 3) put(P + 2 * pstride + off, l);
}

// bit pattern of max |x| (non-negative floats order like unsigned integers); out must be zeroed
__global__ __launch_bounds__(256) void maxabs_kernel(int64_t n, int64_t K, int64_t ld, const float* __restrict__ x, unsigned* __restrict__ out) {
    __shared__ unsigned wm[4];
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned b = __builtin_bit_cast(unsigned, x[(i / K) * ld + i % K]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                      // one atomic per workgroup (same-address atomics serialise: 2048 of them took 27 us)
        for (int w = 1; w < 4; ++w) m = wm[w] > m ? wm[w] : m;
        atomicMax(out, m);
    }
}

// the same for a contiguous operand (ld == K), 16 bytes per lane and load -- r06: a 512 x 131 072 Gram (the deep GP's first layer) took the
// element-wise form above 0.33 ms (two 64-bit divisions per element, 256 workgroups); this one reads it at the streaming rate
__global__ __launch_bounds__(256) void maxabs_flat_kernel(int64_t n4, const uint4* __restrict__ x, unsigned* __restrict__ out) {
    __shared__ unsigned wm[4];
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 v = x[i];
        const unsigned a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = wm[w] > m ? wm[w] : m;
        atomicMax(out, m);
    }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct SplitArgs {
    const unsigned short* A; const unsigned short* B; float* C;
    int64_t M, N, K16;          // K16 = number of 16-wide k blocks
    int64_t pA, pB, ldc;
    float alpha, beta;
    int splitk, lower_only, atomic, nprod, use_dma;
    int c_blk;                  // C in 16-column blocks: element (row, col) at ((col / 16) * M + row) * 16 + col % 16 (the SVGP reverse pass reads T so)
    int64_t kchunk;             // k blocks per split
    int64_t tm, tn, ntiles, nwg;
    const float* ad0; int pow0;          // alpha *= ad0[0]^pow0 (device scalar, e.g. the kernel variance of Gram planes)
    const unsigned* maxbits;             // alpha /= scale_from_maxbits(maxbits[0]) (the power-of-two scale of an f16x2 operand)
    const unsigned* maxbits2;            // the same for the other operand
    // rendezvous of the workgroups that share operand lines (wide kernels; see wg_rendezvous): counters (nullptr = none), workgroups per
    // group, loop trips between two rendezvous (0 = one per work item, at its start: group = sync_n consecutive work items), counters per k split
    unsigned* sync; int sync_n, sync_period, sync_slots;
    unsigned* maxout;                    // wide kernels, plain (beta = 0, unsplit) products: atomicMax of the bit pattern of max |C| (nullptr: none)
    // c_blk == 2 (wide kernels, CPL instantiations): C is written as TWO f16 PLANES (hi + lo of alpha * acc) in the 16-column-block order --
    // i.e. directly as the (row, k = col) operand of a following split product (the whitened SVGP tier: V = L^-1 Kuf feeds Phi = V V^T)
    unsigned short* Cp; int64_t pC;
    int64_t rot_div;
    int a_lower;                         // A is lower triangular (A[m][k] = 0 for k > m): a row tile's k loop ends at its last row
    // a_lower, CPL instantiations (r05): work items are PAIRS over two neighbouring column strips -- role r takes row tile r of the first
    // strip with k ASCENDING, then row tile tm - 1 - r of the second strip with k DESCENDING: every role carries tm + 1 units of k blocks,
    // and when the tm roles of a pair start together (rendezvous) every one of them reads k block j of the first strip at time j and k
    // block j of the second strip at time (tm + 1) units - j -- the strip's B planes are fetched from the fabric once instead of once per
    // row tile (r04 PMC: 21.6 GB for an 8.6 GB operand = (1 + 2 + 3 + 4) / 4)
    int pair;
    // CPL instantiations, optional: the same values ALSO as the planes of the transposed (N x K' = M) operand, element (n, m) at
    // ((m / 16) * N + n) * 16 + m % 16 (through a wave-private LDS tile, 16-byte stores), and with avec the partial sums
    // Upart[(m / 128) * N + n] = sum over the 128-row band of avec[m] * (hi + lo)(m, n)  (the whitened SVGP tier's V^T and a^T V)
    unsigned short* Ct; int64_t pCt;
    const float* avec; float* Upart;
    int cp_nt;                           // bit 0: the planes, bit 1: the transposed planes are stored non-temporally
    mxf_fuse_args fz;                    // FUSE instantiation: the SVGP reverse pass in the epilogue (internal.h)
};

__device__ __forceinline__ int lds_unit(int row, int kh) { return row * 2 + (kh ^ ((row >> 3) & 1)); }

template <bool DMA, int NP>
__global__ __launch_bounds__(SNT, (NP == 2 && !DMA) ? 4 : 3) void gemm_split_kernel(SplitArgs g) {
    // [stage][A|B][plane][unit]: two stages of 24 KB for three planes; THREE stages of 16 KB for the two-plane LDS-DMA kernel, whose
    // loads run two k blocks ahead of the MFMAs (one block ahead left the matrix pipe waiting for L2: 14.5 ms at the T shape, of which
    // only 7 ms scale with the number of products)
    constexpr int NST = (DMA && NP == 2) ? 3 : 2;
    __shared__ u32x4 smem[NST][2][NP][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    int64_t wid = blockIdx.x;
    {   // XCD-aware mapping (see gemm.hip)
        const int64_t q = g.nwg / 8, r = g.nwg % 8, xcd = wid % 8, j = wid / 8;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int64_t split = wid / g.ntiles;
    int64_t t = wid % g.ntiles;
    int64_t tile_m, tile_n;
    if (g.lower_only) {
        int64_t row = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        tile_m = row; tile_n = t - row * (row + 1) / 2;
    } else {
        tile_m = t % g.tm; tile_n = t / g.tm;
    }
    const int64_t m0 = tile_m * SBM, n0 = tile_n * SBN;
    const int64_t kbeg = split * g.kchunk;
    const int64_t kend = (kbeg + g.kchunk < g.K16) ? kbeg + g.kchunk : g.K16;

    f32x16 c[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[x][y][r] = 0.f;

    // loader: thread t <-> 16-byte unit t of the (128 x 16) slab (row = t >> 1, k half = t & 1)
    const int lrow = tid >> 1, lkh = tid & 1;
    const bool va = (m0 + lrow) < g.M, vb = (n0 + lrow) < g.N;
    const unsigned short* pa = g.A + (m0 + lrow) * 16 + lkh * 8;
    const unsigned short* pb = g.B + (n0 + lrow) * 16 + lkh * 8;
    const int sunit = lds_unit(lrow, lkh);
    u32x4 ra[NP], rb[NP];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#define SLOAD(kb)                                                                                                                  \
    do {                                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                           \
            ra[p] = va ? *reinterpret_cast<const u32x4*>(pa + p * g.pA + (kb) * g.M * 16) : zero4;                                 \
            rb[p] = vb ? *reinterpret_cast<const u32x4*>(pb + p * g.pB + (kb) * g.N * 16) : zero4;                                 \
        }                                                                                                                          \
    } while (0)
#define SSTORE(buf)                                                                                                                \
    do {                                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) { smem[buf][0][p][sunit] = ra[p]; smem[buf][1][p][sunit] = rb[p]; }         \
    } while (0)
    // interior tiles: LDS-DMA (global_load_lds_dwordx4) straight into the other LDS buffer -- no staging VGPRs, no ds_write pass.
    // The LDS destination of a wave is linear (base + lane * 16), so the XOR swizzle is applied to the SOURCE: the lane that fills
    // LDS unit u = (row, khs) fetches global k half khs ^ ((row >> 3) & 1) of that row.
    constexpr bool dma = DMA;     // host guarantees M, N multiples of 128 for the DMA instantiation
    const int drow = tid >> 1, dkh = (tid & 1) ^ ((drow >> 3) & 1);
    const unsigned short* da = g.A + (m0 + drow) * 16 + dkh * 8;
    const unsigned short* db = g.B + (n0 + drow) * 16 + dkh * 8;
#define SDMA(kb, buf)                                                                                                              \
    do {                                                                                                                           \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                                           \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + p * g.pA + (kb) * g.M * 16),     \
                                             (__attribute__((address_space(3))) void*)(&smem[buf][0][p][wave * 64]), 16, 0, 0);    \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + p * g.pB + (kb) * g.N * 16),     \
                                             (__attribute__((address_space(3))) void*)(&smem[buf][1][p][wave * 64]), 16, 0, 0);    \
        }                                                                                                                          \
    } while (0)
    if (kbeg < kend) {
        if constexpr (dma) SDMA(kbeg, 0);
        else { SLOAD(kbeg); SSTORE(0); }
    }
    if constexpr (NST == 3) {
        if (kbeg + 1 < kend) {
            SDMA(kbeg + 1, 1);
            __builtin_amdgcn_s_waitcnt(0xF70 | (2 * NP));       // vmcnt(2 NP): everything but the stage just issued has landed
        } else {
            __builtin_amdgcn_s_waitcnt(0xF70);                  // vmcnt(0)
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }
    const int li = lane & 31, lk = lane >> 5;
    int ua[2], ub[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) { ua[x] = lds_unit(wm + 32 * x + li, lk); ub[x] = lds_unit(wn + 32 * x + li, lk); }
    int cur = 0;
    for (int64_t kb = kbeg; kb < kend; ++kb) {
        const bool more = kb + 1 < kend;
        if constexpr (NST == 3) {
            if (kb + 2 < kend) SDMA(kb + 2, (cur + 2) % 3);
        } else {
            if (more) { if constexpr (dma) SDMA(kb + 1, cur ^ 1); else SLOAD(kb + 1); }
        }
        u32x4 a[2][NP], b[2][NP];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                a[x][p] = smem[cur][0][p][ua[x]];
                b[x][p] = smem[cur][1][p][ub[x]];
            }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                f32x16 acc = c[x][y];
                if constexpr (NP == 3) {
#define BF(v) __builtin_bit_cast(bf16x8, v)
                    if (g.nprod >= 6) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][1]), BF(b[y][1]), acc, 0, 0, 0);   // m m'
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][0]), BF(b[y][2]), acc, 0, 0, 0);   // h l'
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][2]), BF(b[y][0]), acc, 0, 0, 0);   // l h'
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][0]), BF(b[y][1]), acc, 0, 0, 0);       // h m'
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][1]), BF(b[y][0]), acc, 0, 0, 0);       // m h'
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(a[x][0]), BF(b[y][0]), acc, 0, 0, 0);       // h h'
#undef BF
                } else {
#define HF(v) __builtin_bit_cast(f16x8, v)
                    if (g.nprod >= 3) {      // (MXF_SPLIT_NPROD=1: diagnostic only -- the kernel's time without the cross products)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(a[x][0]), HF(b[y][1]), acc, 0, 0, 0);    // hi lo'
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(a[x][1]), HF(b[y][0]), acc, 0, 0, 0);    // lo hi'
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(a[x][0]), HF(b[y][0]), acc, 0, 0, 0);        // hi hi'
#undef HF
                }
                c[x][y] = acc;
            }
        if constexpr (NST == 3) {
            // the next block's stage must have landed (this wave's share; the barrier extends that to the workgroup); the stage issued
            // in this iteration (2 NP loads per lane) may stay in flight.  A bare s_barrier: __syncthreads() would wait for vmcnt(0).
            asm volatile("" ::: "memory");
            if (kb + 2 < kend) __builtin_amdgcn_s_waitcnt(0xF70 | (2 * NP));
            else __builtin_amdgcn_s_waitcnt(0xF70);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur = cur == 2 ? 0 : cur + 1;
        } else {
            if constexpr (!dma) { if (more) SSTORE(cur ^ 1); }
            __syncthreads();
            cur ^= 1;
        }
    }
#undef SLOAD
#undef SSTORE
#undef SDMA
    float alpha = g.alpha;
    const float beta = g.beta;
    if (g.ad0) { const float v = g.ad0[0]; for (int i = 0; i < g.pow0; ++i) alpha *= v; }
    if (g.maxbits) alpha /= scale_from_maxbits(g.maxbits[0]);
    if (g.maxbits2) alpha /= scale_from_maxbits(g.maxbits2[0]);
    const bool atomic = g.atomic != 0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t col = n0 + wn + y * 32 + (lane & 31);
                if (row < g.M && col < g.N && !(g.lower_only && col > row)) {
                    float* p = g.c_blk ? g.C + ((col >> 4) * g.M + row) * 16 + (col & 15) : g.C + row * g.ldc + col;
                    const float v = alpha * c[x][y][r];
                    if (atomic) atomic_add(p, v);
                    else *p = (beta == 0.f) ? v : v + beta * (*p);
                }
            }
}

// ------------------------------------------------------------------------------------------------ the (32 XT) x 256 kernels (f16x2 operands)
// The 128 x 128 kernel above moves every operand through LDS: per 16-wide k block and workgroup 16 KB of LDS-DMA writes and 32 KB of
// fragment reads for 48 MFMAs, and the LDS (not the matrix pipe: 58 % busy) is what it runs out of.  This kernel cuts the LDS traffic per
// MFMA to a quarter:
//   * block tile (32 XT) (A rows) x 256 (B rows), four waves side by side along B: wave w owns ALL A rows x columns [64 w, 64 w + 64)
//     = XT x 2 MFMA tiles, 6 XT v_mfma_f32_32x32x16_f16 per k block.  XT = 4: 128 rows, 128 accumulator registers, two workgroups per CU;
//     XT = 8: 256 rows, 256 accumulators (the whole AGPR half of the register file), one workgroup per CU -- per MFMA it moves 2/3 of
//     the bytes the 128-row tile moves from L2 (A 16 KB + B 16 KB per 192 MFMAs instead of 8 + 16 per 96);
//   * A (shared by the four waves) goes through LDS: LDS-DMA, a three-slot ring, 2 XT ds_read_b128 per wave and k block;
//   * B is private to a wave, so it never touches LDS: each lane fetches its 16-byte fragment units straight from global memory
//     (the plane layout makes a wave's 32 rows x 16 k one contiguous 1 KB run) two k blocks ahead into a three-deep register ring.
// One barrier per k block.  Inline-asm loads + hand-counted s_waitcnt vmcnt: the compiler's own waits would drain the LDS-DMA queue at the
// first use of an ordinary load (cdna_hip_programming.md section 5, "mixing load kinds").  The MFMA operands are swapped (D = B A^T), which
// leaves every lane with four CONSECUTIVE output columns per accumulator quad: the epilogue is 16-byte stores.
//
// L2 locality by rendezvous (r03).  The workgroups that share operand lines -- the row tiles of one column strip of T (same B strip, and
// with the whole XCD in step also the same A k-window), the tiles of one k split of Psi2 -- are dealt to ONE XCD, but nothing kept them in
// step: each walks its own k loop, they drift apart by more than the 4 MB L2 holds, and every operand line was fetched ~3 times from the
// fabric (r02 PMC: 26.2 GB for 8.6 GB of planes).  wg_rendezvous() is a BOUNDED spin on one counter per group (at the start of a work item
// for T, every `sync_period` loop trips for the long-K products): a pacing hint, never needed for correctness -- a workgroup that waited
// SYNC_LIMIT for its partners goes on alone and stops waiting after its second time-out, so a launch next to kernels that hold some CUs,
// or two such launches on different streams, cannot deadlock.  Every workgroup adds exactly 2 to its group's counter (arrive + depart,
// or both at once when it no longer waits); the add that completes 2 n resets the word, so the counters are zero between launches.
constexpr int WBN = 256;
constexpr unsigned long long SYNC_LIMIT = 2000ull;       // wall_clock64 ticks (100 MHz): 20 us

__device__ __forceinline__ void wg_rendezvous(unsigned* ctr, unsigned n, int& patience) {
    if (threadIdx.x == 0) {
        if (patience > 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
                if (wall_clock64() - t0 > SYNC_LIMIT) { --patience; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == 2u * n)
                __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (__hip_atomic_fetch_add(ctr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2u == 2u * n) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_barrier();        // bare: requests in flight (LDS-DMA, the previous item's stores) stay in flight
}

__device__ __forceinline__ u32x4 gload16(unsigned voff, const void* sbase) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
__device__ __forceinline__ u32x4 gload16_o1024(unsigned voff, const void* sbase) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}

// NH = 2: 512-thread workgroups, two row halves of four waves each (wave = 4 h + w owns rows [32 XT h, +32 XT) x columns [64 w, +64)):
// the 256-row tile with TWO waves per SIMD -- a wave's loads, LDS reads and waits run under its partner's MFMAs, which a single
// 512-register wave per SIMD cannot do (r03: the XT = 8, NH = 1 form halves the fabric fetch and clocks 1.83 instead of 1.56 GHz, but its
// matrix pipe idles through every ds_read / VMEM issue: same wall time).  The two halves share the A slab in LDS; both fetch the B
// fragments of their column quarter (identical addresses, a few hundred cycles apart: the second is an L1 / L2 hit, no fabric traffic).
//
// PP ("ping-pong", NH = 2 only): the two row halves run half a k step apart.  A k step is split into a LOAD phase (request block k + 2,
// read the A fragments of block k from LDS, wait for block k + 1) and a COMPUTE phase (24 MFMAs), with a workgroup barrier after each; the
// second half starts one phase late, so on every SIMD one wave computes while its partner loads.  Without it both waves of a SIMD reach the
// shared barrier together, want the matrix pipe together and then wait for LDS / memory together.
// BLO = false: the B operand enters through its HIGH plane only (B rounded to f16, A still exact: two products instead of three, half the B
// fetch) -- for products that feed GRADIENTS only (the T of the SVGP step; DESIGN.md section 4 states the error this leaves in them).
// LSKIP (lower-only products): a wave whose 128 x 64 block lies strictly ABOVE the diagonal (the upper right quarter of a diagonal 256 x 256
// tile: 2 of its 8 waves) runs a k loop WITHOUT its B loads and MFMAs -- it still fetches its share of the A slab and keeps every barrier.
// (A branch around the MFMAs inside the one loop cost the kernel its schedule, see below; here the loop exists twice and the idle form is a
//  separate instantiation, so the T product's kernel is untouched.)  The busiest SIMDs still carry two working waves: the gain is power.
// BFI (probe builds only, MXF_SPLIT_BF16MFMA=1): the SAME kernel with v_mfma_f32_32x32x16_bf16 on the same bits -- the numbers mean nothing,
// the time does: it separates the operand FORMAT (11-bit f16 mantissas vs 8-bit bf16 ones toggling the multiplier array; the guide's MFMA
// microbenchmark gives 2178 vs 2382 TF) from the schedule when this kernel is compared with the guide's bf16 GEMM template.
// FUSE (r05): the product is the T = H0 Kuf (or Hh V) of the SVGP training call and its epilogue IS the reverse pass of that call (see
// mxf_fuse_args, internal.h): nothing of C is written.  Per 32 x 32 accumulator fragment (rows m on lanes, 16 columns n per lane):
//   dots   x_n . z_m in the SAME accumulator layout, four v_mfma_f32_32x32x2_f32 (true float32: they feed r2 of near pairs);
//   k = 2^(esc - r2), u = T + w_m e_n, W = u k  (the RBF weight up to -(c1 variance) 2^-esc, applied when the sums are flushed);
//   row side  [B | S]_m += W [X | 1]: W's accumulator quads, converted to f16 hi + lo and paired by v_permlane32_swap, ARE the A operand
//             (k = n) of v_mfma_f32_32x32x16_f16; the B operand [x_n | 1] (hi / lo) comes from an LDS table built per item;
//   col side  [D | C]_n += W^T [Z | 1]: the same f16 pairs go through a wave-private LDS tile [m][n] and come back TRANSPOSED through
//             ds_read_b64_tr_b16 (lane n, 8 consecutive m) as the A operand (k = m); B = [z_m | 1] from an LDS table built per row tile.
// Row sums accumulate in LDS across all items of the workgroup (a persistent workgroup keeps its row tile) and leave as float64 atomics at
// the end; column sums leave per item as float32 atomics into dX (two row halves x M / 256 row tiles per element).
template <int XT, int NH, bool PP, bool BLO = true, bool LSKIP = false, bool CPL = false, bool BFI = false, bool FUSE = false>
__device__ __forceinline__ void wide_body(const SplitArgs& g) {
    static_assert(!FUSE || (XT == 4 && NH == 2 && !PP && BLO && !LSKIP && !CPL && !BFI), "the fused reverse pass lives in the eight-wave 256 x 256 product");
    static_assert(!PP || NH == 2, "ping-pong needs the two row halves");
    static_assert(!LSKIP || (!PP && BLO), "the idle-wave loop mirrors the plain pipelined loop");
    constexpr int WBMt = 32 * XT * NH;             // A rows per tile
    constexpr int NU = 64 * XT * NH;               // 16-byte units of one plane's (WBMt x 16) slab
    constexpr int ND = XT / 4;                     // LDS-DMA requests per thread, plane and k block
    constexpr int NTH = 256 * NH;
    __shared__ u32x4 smem[3][2][NU];               // [ring slot][plane][16-byte unit]: 3 x 8 KB (128 rows) / 3 x 16 KB (256 rows), A only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wq = wave & 3, wh = wave >> 2;
    int patience = 2;
    float cmax = 0.f;                              // max |C| over this thread's outputs (g.maxout)
    // ---- FUSE: tables and constants of the fused reverse pass -------------------------------------------------------------------------
    // fz_xb / fz_zb [plane][block of 16 k][j = 16 output columns][16 k] halves: B operands of the two accumulating products ([x | 1] of the
    // item's 256 columns, [z | 1] of the row tile's 256 rows; j = 8 is the ones column, j > 8 zero); fz_col [|x_n|^2 - esc | e_n]; fz_row
    // [256 rows][12]: 0..7 B_mq, 8 S_m, 9 R_m in the pass's units
    constexpr int FZN = FUSE ? 2 * 16 * 16 * 16 : 8;
    __shared__ __attribute__((aligned(16))) unsigned short fz_xb[FZN], fz_zb[FZN];
    __shared__ __attribute__((aligned(16))) float fz_col[FUSE ? 2 * 256 : 4];
    // float32 copies for the dot products and the flushes (LDS latency instead of a global round trip per fragment: the first form of this
    // epilogue fetched them from global memory fragment by fragment and spent 97 us per item, mostly waiting): [z (8) | |z|^2 | w | - | -] per row
    // of the row tile, [x (8)] per column of the item
    __shared__ __attribute__((aligned(16))) float fz_zf[FUSE ? 256 * 12 : 4], fz_xf[FUSE ? 256 * 8 : 4];
    __shared__ __attribute__((aligned(16))) float fz_row[FUSE ? 256 * 12 : 4];
    float fz_escf = 0.f, fz_unsc = 1.f, fz_fl = 1.f, fz_var = 1.f, fz_ilj = 0.f, fz_dl3 = 0.f;
    int64_t fz_m0 = -1;
    constexpr float FZ_CS = 0.84932180028801904272f;       // the coordinates carry sqrt(log2(e) / 2): k = 2^-r2 (bwd_prescale_kernel)
    if constexpr (FUSE) {
        const mxf_fuse_args& z = g.fz;
        fz_var = z.var[0];
        const float c1 = (float)z.a1 / z.noise[0];
        // |u k| <= bound: |T| <= max(variance, sqrt(variance)) M max |A operand| (explicit form: Gram planes hold k / variance, alpha carries
        // the variance; whitened: V / sigma), |w e| <= max |w| max |e|.  (The separate pass took max |T| itself from the product: ~2^13 tighter;
        // a loose bound costs absolute precision of the f16 pairs only -- dX 1.6e-6 -> 5e-6 when this bound was first tried, r03.)
        const float vb = fmaxf(fz_var, sqrtf(fz_var));
        const float bnd = vb * (float)g.M * __builtin_bit_cast(float, z.h0max[0]) + __builtin_bit_cast(float, z.mx[0]) * __builtin_bit_cast(float, z.mx[1]);
        const int ex = (int)((__builtin_bit_cast(unsigned, bnd) >> 23) & 0xff);
        int esc = (ex == 0 || ex == 0xff) ? 0 : 13 - (ex - 127);
        esc = esc > 60 ? 60 : (esc < -60 ? -60 : esc);
        fz_escf = (float)esc;
        fz_unsc = __builtin_bit_cast(float, (unsigned)(127 - esc) << 23);
        fz_fl = -c1 * fz_var * fz_unsc;
        const int lj = lane & 31;
        fz_ilj = (lj < z.Q) ? 1.f / z.ls[z.ard ? lj : 0] : 0.f;
        for (int i = tid; i < FZN; i += NTH) {           // zero, and 1.0 in the hi plane's ones column
            const unsigned short one = ((i >> 4) & 15) == 8 && i < FZN / 2 ? (unsigned short)0x3C00 : (unsigned short)0;
            fz_xb[i] = one; fz_zb[i] = one;
        }
        for (int i = tid; i < 256 * 12; i += NTH) fz_row[i] = 0.f;
        __syncthreads();
    }
    // rows of fz_row -> the float64 accumulators (when the workgroup's row tile changes, and at the end)
    auto fz_flush_rows = [&](int64_t m0_) {
        if constexpr (FUSE) {
            __syncthreads();
            for (int i = tid; i < 256 * 10; i += NTH) {
                const int r = i / 10, cc = i % 10;
                const float v = fz_row[r * 12 + cc];
                if (v != 0.f) atomic_add(g.fz.zacc + (m0_ + r) * 16 + cc, (double)v * (double)(cc == 9 ? fz_var * fz_unsc : fz_fl));
                fz_row[r * 12 + cc] = 0.f;
            }
            __syncthreads();
        }
    };
    // Persistent over the (tile, k split) work items: workgroup b takes items b, b + gridDim.x, ... (gridDim.x a multiple of 8, so its items
    // stay on its XCD's run of tiles).  The epilogue's stores of one item drain while the next item's first loads are in flight; with one
    // item per workgroup every tile paid a dispatch + an un-overlapped pipeline fill + a store burst (~30 % of a K = 1024 tile).
    const int nsub = (CPL && g.pair) ? 2 : 1;
    for (int64_t wid0 = blockIdx.x; wid0 < g.nwg; wid0 += gridDim.x)
    for (int sub = 0; sub < nsub; ++sub) {
    int64_t wid = wid0;
    {   // XCD-aware mapping: every XCD owns a contiguous run of tiles
        const int64_t q = g.nwg / 8, r = g.nwg % 8, xcd = wid % 8, j = wid / 8;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int64_t split = wid / g.ntiles;
    int64_t t = wid % g.ntiles;
    int64_t tile_m, tile_n;
    if (g.lower_only) {             // tiles (tm, tn) with 256 tn <= WBMt tm + WBMt - 1: row tm holds (WBMt tm + WBMt - 1) / 256 + 1 tiles
        int64_t row = 0;
        while (t >= (WBMt * row + WBMt - 1) / WBN + 1) { t -= (WBMt * row + WBMt - 1) / WBN + 1; ++row; }
        tile_m = row; tile_n = t;
    } else {
        tile_m = t % g.tm; tile_n = t / g.tm;       // the row tiles of one column tile are neighbours: they share the B columns in L2
        // triangular A: row tile r carries (r + 1) / tm of the work, and a persistent workgroup would meet the SAME row tile in every
        // round (its items are a multiple of tm apart) -- rotate the row tile with the round
        if constexpr (CPL) {
            if (g.pair) {            // item t = (strip pair t / tm, role t % tm); see SplitArgs::pair
                const int64_t role = t % g.tm, sp = t / g.tm;
                tile_n = 2 * sp + sub;
                tile_m = sub ? g.tm - 1 - role : role;
                if (tile_n >= g.tn) break;          // odd strip count: the last pair has no second strip (workgroup-uniform)
            } else if (g.a_lower && g.rot_div > 0) tile_m = (tile_m + tile_n / g.rot_div) % g.tm;
        }
    }
    const int64_t m0 = tile_m * WBMt, n0 = tile_n * WBN;
    const int64_t kbeg = split * g.kchunk;
    int64_t kend = (kbeg + g.kchunk < g.K16) ? kbeg + g.kchunk : g.K16;
    if constexpr (CPL) { if (g.a_lower) { const int64_t kl = (m0 + WBMt + 15) / 16; kend = kl < kend ? kl : kend; } }
    // k order of this item: block KMAP(i), i = 0 .. nk - 1 (descending for the second item of a pair; compile-time ascending otherwise)
    const bool kdesc = CPL && g.pair && sub == 1;
    const int64_t korg = kdesc ? kend - 1 : kbeg, kdir = kdesc ? -1 : 1, nk = kend - kbeg;
#define KMAP(i) (korg + kdir * (i))
    if (g.sync && g.sync_period == 0 && sub == 0) wg_rendezvous(g.sync + wid / g.sync_n, (unsigned)g.sync_n, patience);
    // (Measured and dropped, r03: skipping the MFMAs of the waves / 32 x 32 fragments that lie strictly above the diagonal of a lower-only
    //  product -- 10 % / 17 % of Psi2's matrix work.  Any branch around the MFMA block costs the kernel its schedule: 231 -> 251 VGPRs with
    //  the wave-uniform form, T 11.5 -> 12.5 ms although T never takes the branch; spills with the per-fragment form.)

    f32x16 c[XT][2];
#pragma unroll
    for (int x = 0; x < XT; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[x][y][r] = 0.f;

    // A: LDS-DMA, thread t fills 16-byte units t (+ NTH u) of each plane's slab; the XOR swizzle of the two k halves is applied to the
    // source (unit U <-> row U >> 1; NTH more units = NTH / 2 more rows leave the swizzle bit (row >> 3) & 1 as it is)
    const int drow = tid >> 1, dkh = (tid & 1) ^ ((drow >> 3) & 1);
    const unsigned short* da = g.A + (m0 + drow) * 16 + dkh * 8;
    // B: lane <-> (row = lane & 31, k half = lane >> 5) of the wave's two 32-row fragments; per-lane byte offset inside the k block's slab
    const unsigned bvoff = (unsigned)(((64 * wq + (lane & 31)) * 16 + (lane >> 5) * 8) * 2);
    const unsigned short* bbase = g.B + n0 * 16;                     // wave-uniform
    const int li = lane & 31, lk = lane >> 5;
    int ua[XT];
#pragma unroll
    for (int x = 0; x < XT; ++x) ua[x] = lds_unit(32 * XT * wh + 32 * x + li, lk);

    u32x4 b0[2][2], b1[2][2], b2[2][2];          // register ring of the B fragments [y][plane]: blocks kb, kb + 1, kb + 2
    // per k block 4 + 2 ND requests, in this order: four B fragment loads (registers), 2 ND LDS-DMA requests (A slab)
#define W_ISSUE(kb, SLOT, BR)                                                                                                       \
    do {                                                                                                                            \
        const unsigned short* bk_ = bbase + (kb) * g.N * 16;                                                                        \
        BR[0][0] = gload16(bvoff, bk_);                                                                                             \
        BR[1][0] = gload16_o1024(bvoff, bk_);                                                                                       \
        if constexpr (BLO) {                                                                                                        \
            BR[0][1] = gload16(bvoff, bk_ + g.pB);                                                                                  \
            BR[1][1] = gload16_o1024(bvoff, bk_ + g.pB);                                                                            \
        }                                                                                                                           \
        _Pragma("unroll") for (int u_ = 0; u_ < ND; ++u_) {                                                                         \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + u_ * (NTH / 2) * 16 + (kb) * g.M * 16), \
                                             (__attribute__((address_space(3))) void*)(&smem[SLOT][0][wave * 64 + NTH * u_]), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + u_ * (NTH / 2) * 16 + g.pA + (kb) * g.M * 16), \
                                             (__attribute__((address_space(3))) void*)(&smem[SLOT][1][wave * 64 + NTH * u_]), 16, 0, 0); \
        }                                                                                                                           \
    } while (0)
    // all but the requests of the block issued last have landed (this wave's share; the barrier extends it to the workgroup).  The B
    // registers that are now complete are tied to the wait ("+v"), so that no use of them can be scheduled above it.  The wait always sits
    // in STRAIGHT-LINE code: inside a branch the compiler may place the register copies of a control-flow merge in front of it, i.e. copy
    // registers whose loads are still in flight (seen in an earlier form of this kernel; csrc/check_wide_isa.py guards against it).
#define W_WAIT_ASM(NSTR, BR)                                                                                                        \
    do {                                                                                                                            \
        if constexpr (BLO) asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(BR[0][0]), "+v"(BR[0][1]), "+v"(BR[1][0]), "+v"(BR[1][1])::"memory"); \
        else asm volatile("s_waitcnt vmcnt(" NSTR ")" : "+v"(BR[0][0]), "+v"(BR[1][0])::"memory");                                  \
    } while (0)
    // requests per k block: the B fragment loads (4, or 2 without the low plane) + 2 ND LDS-DMA requests
#define W_WAIT1(BR)                                                                                                                 \
    do {                                                                                                                            \
        if constexpr (XT == 4 && BLO) W_WAIT_ASM("6", BR); else if constexpr (XT == 4) W_WAIT_ASM("4", BR);                         \
        else if constexpr (BLO) W_WAIT_ASM("8", BR); else W_WAIT_ASM("6", BR);                                                      \
        __builtin_amdgcn_s_barrier();                                                                                               \
        asm volatile("" ::: "memory");                                                                                              \
    } while (0)
#define W_WAIT0(BR)                                                                                                                 \
    do {                                                                                                                            \
        W_WAIT_ASM("0", BR);                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                               \
        asm volatile("" ::: "memory");                                                                                              \
    } while (0)
#define HF(v) __builtin_bit_cast(f16x8, v)
#define WMMA(a, b, c) (BFI ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0) \
                           : __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(a), HF(b), c, 0, 0, 0))
#define W_COMPUTE(SLOT, BR)                                                                                                         \
    do {                                                                                                                            \
        u32x4 a_[XT][2];                                                                                                            \
        _Pragma("unroll") for (int x = 0; x < XT; ++x) { a_[x][0] = smem[SLOT][0][ua[x]]; a_[x][1] = smem[SLOT][1][ua[x]]; }         \
        W_MFMAS(a_, BR);                                                                                                            \
    } while (0)
#define W_MFMAS(a_, BR)                                                                                                             \
    do {                                                                                                                            \
        _Pragma("unroll") for (int x = 0; x < XT; ++x)                                                                              \
            _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                           \
                c[x][y] = WMMA(BR[y][0], a_[x][1], c[x][y]);                                                         /* hi' lo */   \
        if constexpr (BLO) {                                                                                                        \
        _Pragma("unroll") for (int x = 0; x < XT; ++x)                                                                              \
            _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                           \
                c[x][y] = WMMA(BR[y][1], a_[x][0], c[x][y]);                                                         /* lo' hi */   \
        }                                                                                                                           \
        _Pragma("unroll") for (int x = 0; x < XT; ++x)                                                                              \
            _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                           \
                c[x][y] = WMMA(BR[y][0], a_[x][0], c[x][y]);                                                         /* hi' hi */   \
    } while (0)
    // one k block `kk`: ring slot SLOT / registers BR hold it; block kk + 2 (clamped to the last block: the loop is branch-free, the
    // surplus requests of the last two steps re-read the last block and are never used) is requested into SLOT2 / BR2 -- free since the
    // barrier that closed block kk - 1 --, then the MFMAs, then the wait for block kk + 1 (registers BRN) with block kk + 2 in flight
#define W_STEP(kk, SLOT, BR, SLOT2, BR2, BRN)                                                                                       \
    do {                                                                                                                            \
        const int64_t i2_ = (kk) + 2 < nk ? (kk) + 2 : nk - 1;                                                                      \
        const int64_t k2_ = KMAP(i2_);                                                                                              \
        W_ISSUE(k2_, SLOT2, BR2);                                                                                                   \
        W_COMPUTE(SLOT, BR);                                                                                                        \
        W_WAIT1(BRN);                                                                                                               \
    } while (0)
    // ping-pong form of a step: LOAD phase | barrier | COMPUTE phase | barrier.  Nothing may be scheduled across the barriers (the MFMAs
    // are not memory operations: only sched_barrier keeps them on their side).  The fragment reads are complete (lgkmcnt(0)) before the
    // barrier that ends the load phase: the slot they read is rewritten by a partner's request two barriers later.
#define W_PHASE_END()                                                                                                               \
    do {                                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    } while (0)
#define W_STEP_PP(kk, SLOT, BR, SLOT2, BR2, BRN)                                                                                    \
    do {                                                                                                                            \
        const int64_t i2_ = (kk) + 2 < nk ? (kk) + 2 : nk - 1;                                                                      \
        const int64_t k2_ = KMAP(i2_);                                                                                              \
        W_ISSUE(k2_, SLOT2, BR2);                                                                                                   \
        u32x4 a_[XT][2];                                                                                                            \
        _Pragma("unroll") for (int x = 0; x < XT; ++x) { a_[x][0] = smem[SLOT][0][ua[x]]; a_[x][1] = smem[SLOT][1][ua[x]]; }         \
        if constexpr (BLO) W_WAIT_ASM("6", BRN); else W_WAIT_ASM("4", BRN);                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
        W_PHASE_END();                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        W_MFMAS(a_, BR);                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
        W_PHASE_END();                                                                                                              \
    } while (0)
    bool idle = false;
    if constexpr (LSKIP) idle = g.lower_only && (m0 + 32 * XT * wh + 32 * XT - 1) < (n0 + 64 * wq);      // wave-uniform
    if (LSKIP && idle) {
        if (kbeg < kend) {
            // the same sequence of A requests, barriers and rendezvous as the working waves' loop below
#define W_ISSUE_A(kb, SLOT)                                                                                                         \
            do {                                                                                                                    \
                _Pragma("unroll") for (int u_ = 0; u_ < ND; ++u_) {                                                                 \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + u_ * (NTH / 2) * 16 + (kb) * g.M * 16), \
                                                     (__attribute__((address_space(3))) void*)(&smem[SLOT][0][wave * 64 + NTH * u_]), 16, 0, 0); \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da + u_ * (NTH / 2) * 16 + g.pA + (kb) * g.M * 16), \
                                                     (__attribute__((address_space(3))) void*)(&smem[SLOT][1][wave * 64 + NTH * u_]), 16, 0, 0); \
                }                                                                                                                   \
            } while (0)
#define W_WAITL(NSTR) do { asm volatile("s_waitcnt vmcnt(" NSTR ")" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
            const int64_t klast = kend - 1;
            int64_t kb = kbeg;
            for (int i = (int)((kend - kbeg) % 3); i > 0; --i, ++kb) {
                W_ISSUE_A(kb, 0);
                W_WAITL("0");
                __builtin_amdgcn_s_barrier();
            }
            if (kb < kend) {
                W_ISSUE_A(kb, 0);
                W_ISSUE_A(kb + 1, 1);
                if constexpr (XT == 4) W_WAITL("2"); else W_WAITL("4");
                int trip = 0, sync_ix = 0;
                for (; kb < kend; kb += 3) {
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const int64_t k2_ = kb + s3 + 2 < kend ? kb + s3 + 2 : klast;
                        if (s3 == 0) W_ISSUE_A(k2_, 2); else if (s3 == 1) W_ISSUE_A(k2_, 0); else W_ISSUE_A(k2_, 1);
                        if constexpr (XT == 4) W_WAITL("2"); else W_WAITL("4");
                    }
                    if (g.sync_period > 0 && ++trip == g.sync_period) {
                        trip = 0;
                        if (sync_ix < g.sync_slots) wg_rendezvous(g.sync + split * g.sync_slots + sync_ix, (unsigned)g.sync_n, patience);
                        ++sync_ix;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#undef W_WAITL
#undef W_ISSUE_A
        }
    } else
    if (nk > 0) {
        int64_t kb = 0;                            // index into this item's k order (KMAP)
        // the block count modulo 3 first, unpipelined (request, drain, multiply): the pipelined loop then runs whole trips of three
        for (int i = (int)(nk % 3); i > 0; --i, ++kb) {
            W_ISSUE(KMAP(kb), 0, b0);
            W_WAIT0(b0);
            W_COMPUTE(0, b0);
            __builtin_amdgcn_s_barrier();          // slot 0 is rewritten by the next request
        }
        if (kb < nk) {
            W_ISSUE(KMAP(kb), 0, b0);
            W_ISSUE(KMAP(kb + 1), 1, b1);
            W_WAIT1(b0);
            if constexpr (PP) { if (wh == 1) W_PHASE_END(); }         // the second half runs one phase behind from here on ...
            int trip = 0, sync_ix = 0;
            for (; kb < nk; kb += 3) {             // three k blocks per trip: ring indices are compile-time constants, one loop exit
                if constexpr (PP) {
                    W_STEP_PP(kb, 0, b0, 2, b2, b1);
                    W_STEP_PP(kb + 1, 1, b1, 0, b0, b2);
                    W_STEP_PP(kb + 2, 2, b2, 1, b1, b0);
                } else {
                W_STEP(kb, 0, b0, 2, b2, b1);
                W_STEP(kb + 1, 1, b1, 0, b0, b2);
                W_STEP(kb + 2, 2, b2, 1, b1, b0);
                }
                // long-K products: every sync_period trips the tiles of this k split wait for each other (bounded), so that the operand
                // rows they share are fetched into the XCD's L2 once.  (The branch defines none of the ring registers: no merge copies.)
                if (g.sync_period > 0 && ++trip == g.sync_period) {
                    trip = 0;
                    if (sync_ix < g.sync_slots) wg_rendezvous(g.sync + split * g.sync_slots + sync_ix, (unsigned)g.sync_n, patience);
                    ++sync_ix;
                }
            }
            if constexpr (PP) { if (wh == 0) W_PHASE_END(); }         // ... to here: every wave has passed the same number of barriers
            // the surplus requests of the last two steps target registers / LDS the epilogue does not read, but they must have landed
            // before the registers are reused
            if constexpr (BLO)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(b0[0][0]), "+v"(b0[0][1]), "+v"(b0[1][0]), "+v"(b0[1][1]), "+v"(b1[0][0]), "+v"(b1[0][1]),
                             "+v"(b1[1][0]), "+v"(b1[1][1])::"memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(b0[0][0]), "+v"(b0[1][0]), "+v"(b1[0][0]), "+v"(b1[1][0])::"memory");
        }
    }
#undef KMAP
#undef W_STEP
#undef W_STEP_PP
#undef W_PHASE_END
#undef W_MFMAS
#undef WMMA
#undef HF
#undef W_WAIT1
#undef W_WAIT0
#undef W_WAIT_ASM
#undef W_ISSUE
    float alpha = g.alpha;
    const float beta = g.beta;
    if (g.ad0) { const float v = g.ad0[0]; for (int i = 0; i < g.pow0; ++i) alpha *= v; }
    if (g.maxbits) alpha /= scale_from_maxbits(g.maxbits[0]);
    if (g.maxbits2) alpha /= scale_from_maxbits(g.maxbits2[0]);
    const bool atomic = g.atomic != 0;
    // D = B A^T: accumulator register r of tile (x, y) is C[m0 + 32 XT wh + 32 x + (lane & 31)][n0 + 64 wq + 32 y + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)]
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if constexpr (FUSE) {
        const mxf_fuse_args& z = g.fz;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        constexpr int RS = 40;                                   // row stride (halves) of the wave-private transposition tile [32 m][32 n]
        // every wave has left the k loop: the A ring is dead until the next item's first request
        asm volatile("; mxf_fz_epilogue_begin" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        if (m0 != fz_m0) {                                       // (a persistent workgroup keeps its row tile: once per launch)
            if (fz_m0 >= 0) fz_flush_rows(fz_m0);
            fz_m0 = m0;
            const int mm = tid >> 1, hq = tid & 1;               // row mm of the tile, coordinates 4 hq .. + 3
            const f32x4 zv = *reinterpret_cast<const f32x4*>(z.Zs + (m0 + mm) * 8 + 4 * hq);
            *reinterpret_cast<f32x4*>(&fz_zf[mm * 12 + 4 * hq]) = zv;
            if (hq == 0) { fz_zf[mm * 12 + 8] = z.Zn[m0 + mm]; fz_zf[mm * 12 + 9] = z.w[m0 + mm]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 fh = (_Float16)zv[e];
                const _Float16 fo = (_Float16)(zv[e] - (float)fh);
                const int ix = (((mm >> 4) * 16) + 4 * hq + e) * 16 + (mm & 15);
                fz_zb[ix] = __builtin_bit_cast(unsigned short, fh);
                fz_zb[FZN / 2 + ix] = __builtin_bit_cast(unsigned short, fo);
            }
        }
        const int64_t smp = n0 / z.B;                            // B % 256 == 0: the item's columns lie in one sample
        {   // the item's column tables
            const int nn = tid >> 1, hq = tid & 1;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(z.Xs + (n0 + nn) * 8 + 4 * hq);
            *reinterpret_cast<f32x4*>(&fz_xf[nn * 8 + 4 * hq]) = xv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 fh = (_Float16)xv[e];
                const _Float16 fo = (_Float16)(xv[e] - (float)fh);
                const int ix = (((nn >> 4) * 16) + 4 * hq + e) * 16 + (nn & 15);
                fz_xb[ix] = __builtin_bit_cast(unsigned short, fh);
                fz_xb[FZN / 2 + ix] = __builtin_bit_cast(unsigned short, fo);
            }
            if (hq == 0) {
                fz_col[nn] = z.Xn[n0 + nn] - fz_escf;
                fz_col[256 + nn] = z.Y[smp * z.sY + (n0 + nn - smp * z.B)] - z.U[n0 + nn];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
        unsigned short* const wt = reinterpret_cast<unsigned short*>(&smem[0][0][0]) + wave * (2 * 32 * RS);      // [plane][32 m][RS]
        const int g16 = lane >> 4, p16 = lane & 15;              // transposing read: 16-lane group (n block of 16 = g16 & 1, k half = g16 >> 1)
        float qn = 0.f;
        // (column fragments outside, row fragments inside: ONE column-side accumulator is live across the four row fragments, the row-side
        //  sums of a fragment leave for LDS at once -- the 128 accumulators of T leave little room: the other order spilled 209 registers)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int nf = 64 * wq + 32 * y;                     // first column of this fragment column inside the tile
            const f32x4 xf = *reinterpret_cast<const f32x4*>(&fz_xf[(nf + li) * 8 + 4 * lk]);
            f32x16 colacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) colacc[r] = 0.f;
#pragma unroll
            for (int x = 0; x < XT; ++x) {
                const int ml = 32 * XT * wh + 32 * x;            // first row of this fragment inside the tile
                const f32x4 zf = *reinterpret_cast<const f32x4*>(&fz_zf[(ml + li) * 12 + 4 * lk]);
                const f32x2 zw2 = *reinterpret_cast<const f32x2*>(&fz_zf[(ml + li) * 12 + 8]);
                const float zzm = zw2[0], wmm = zw2[1];
                f32x16 dots;
#pragma unroll
                for (int r = 0; r < 16; ++r) dots[r] = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) dots = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[e], zf[e], dots, 0, 0, 0);      // [i = n][j = m], q = 4 lk + e
                unsigned hi[8], lo[8];
                float racc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 xn4 = *reinterpret_cast<const f32x4*>(&fz_col[nf + 8 * q + 4 * lk]);
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(&fz_col[256 + nf + 8 * q + 4 * lk]);
                    float wv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float r2 = fmaf(-2.f, dots[4 * q + t], zzm + xn4[t]);
                        const float k = __builtin_amdgcn_exp2f(-r2);                     // k 2^esc
                        const float tv = alpha * c[x][y][4 * q + t];
                        const float u = fmaf(wmm, e4[t], tv);
                        wv[t] = u * k;
                        qn = fmaf(k, tv, qn);
                        racc = fmaf(k, e4[t], racc);
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const f32x2 v = {wv[2 * d], wv[2 * d + 1]};
                        const f16x2 fh = __builtin_convertvector(v, f16x2);
                        const f16x2 fo = __builtin_convertvector(v - __builtin_convertvector(fh, f32x2), f16x2);
                        hi[2 * q + d] = __builtin_bit_cast(unsigned, fh); lo[2 * q + d] = __builtin_bit_cast(unsigned, fo);
                    }
                }
                // quads (0, 1) and (2, 3): afterwards lane lk owns the EIGHT consecutive columns 16 c + 8 lk .. + 7 of k block c in
                // {hi[4 c], .., hi[4 c + 3]} -- the A operand (k = n) of the row-side product
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hi[4 * cb + d]), "+v"(hi[4 * cb + 2 + d]));
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo[4 * cb + d]), "+v"(lo[4 * cb + 2 + d]));
                    }
                f32x16 rowf;
#pragma unroll
                for (int r = 0; r < 16; ++r) rowf[r] = 0.f;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const u32x4 ah = {hi[4 * cb], hi[4 * cb + 1], hi[4 * cb + 2], hi[4 * cb + 3]}, al = {lo[4 * cb], lo[4 * cb + 1], lo[4 * cb + 2], lo[4 * cb + 3]};
                    const int bix = ((((nf >> 4) + cb) * 16) + (li & 15)) * 16 + 8 * lk;
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(&fz_xb[bix]), bl = *reinterpret_cast<const u32x4*>(&fz_xb[FZN / 2 + bix]);
                    rowf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), rowf, 0, 0, 0);
                    rowf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), rowf, 0, 0, 0);
                    rowf = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), rowf, 0, 0, 0);
                    // the same eight columns of row li into the transposition tile [m = li][n = 16 cb + 8 lk ..]
                    *reinterpret_cast<u32x4*>(wt + li * RS + 16 * cb + 8 * lk) = ah;
                    *reinterpret_cast<u32x4*>(wt + 32 * RS + li * RS + 16 * cb + 8 * lk) = al;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // column side: k = m.  16-lane group g16 = (n block of 16, k half): lane p16 fetches the 4 halves tile[m = 8 kh + 4 rr + p16 / 4]
                // [n = 16 nb + 4 (p16 % 4) ..] and receives tile[m = 8 kh + 4 rr + 0 .. 3][n = 16 nb + p16] (ds_read_b64_tr_b16)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {                 // k blocks of 16 rows m
                    u32x4 th, tl;
                    {
                        const unsigned short* ph = wt + (16 * cc + 8 * (g16 >> 1) + (p16 >> 2)) * RS + 16 * (g16 & 1) + 4 * (p16 & 3);
                        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ph));
                        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ph + 4 * RS));
                        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ph + 32 * RS));
                        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ph + 32 * RS + 4 * RS));
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        const u32x2 a0 = __builtin_bit_cast(u32x2, h0), a1 = __builtin_bit_cast(u32x2, h1), b0_ = __builtin_bit_cast(u32x2, l0), b1_ = __builtin_bit_cast(u32x2, l1);
                        th = u32x4{a0[0], a0[1], a1[0], a1[1]}; tl = u32x4{b0_[0], b0_[1], b1_[0], b1_[1]};
                    }
                    const int zix = ((((ml >> 4) + cc) * 16) + (li & 15)) * 16 + 8 * lk;
                    const u32x4 zh = *reinterpret_cast<const u32x4*>(&fz_zb[zix]), zl = *reinterpret_cast<const u32x4*>(&fz_zb[FZN / 2 + zix]);
                    colacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, tl), __builtin_bit_cast(f16x8, zh), colacc, 0, 0, 0);
                    colacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, th), __builtin_bit_cast(f16x8, zl), colacc, 0, 0, 0);
                    colacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, th), __builtin_bit_cast(f16x8, zh), colacc, 0, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // row side of this fragment: rowf[r] = [B | S] of row ml + 8 (r >> 2) + 4 lk + (r & 3), entry j = li
                if (li < 9) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) __hip_atomic_fetch_add(&fz_row[(ml + 8 * (r >> 2) + 4 * lk + (r & 3)) * 12 + li], rowf[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                racc += __shfl_xor(racc, 32, 64);
                if (lk == 0) __hip_atomic_fetch_add(&fz_row[(ml + li) * 12 + 9], racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_sched_barrier(0);       // keep the unrolled fragments apart: hoisting the next fragment's reads costs registers that are not there
            }
            // column side: colacc[r] = [D | C] of column nf + 8 (r >> 2) + 4 lk + (r & 3), entry j = li (this wave's 128 rows)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float Cn = __shfl(colacc[r], (lane & 32) | 8, 64);
                const int nl = nf + 8 * (r >> 2) + 4 * lk + (r & 3);
                if (li < z.Q) {
                    const float xq = fz_xf[nl * 8 + li];
                    const float pr = xq * Cn;
                    fz_dl3 = fmaf(xq, pr, fz_dl3);
                    if (z.dX) atomic_add(z.dX + (n0 + nl) * z.Q + li, (pr - colacc[r]) * (fz_ilj * (1.f / FZ_CS) * fz_fl));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            double qs = (double)(qn * (fz_var * fz_unsc));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) qs += __shfl_xor(qs, o, 64);
            if (lane == 0) atomic_add(z.scal + 2 * smp, qs);
        }
        // the tables and the transposition tiles are rewritten by the next item (its first A request lands in the ring)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("; mxf_fz_epilogue_end" ::: "memory");
    } else if constexpr (CPL) {
        // planes output: lane (li, lk) holds columns 8 q + 4 lk .. + 3 of quad q.  v_permlane32_swap trades quad q + 1 of the lanes lk = 0 for
        // quad q of the lanes lk = 1: afterwards lane lk owns EIGHT consecutive columns 8 (q + lk) .. + 7 (q even) = one 16-byte unit per
        // plane, and a store instruction covers 32 rows x 32 bytes = 1 KB contiguous of the 16-column block.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        // a of this tile's rows, fetched BEFORE the plane stores are issued and parked in LDS behind the wave tiles: a vector load later in
        // the epilogue waits for vmcnt(0), i.e. for this item's stores to retire (16 such stalls per item cost 1.8 ms of the whitened step)
        float* alds = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(&smem[0][0][0]) + (NTH / 64) * (32 * 68));
        float aval = 0.f;
        if (g.Ct && g.avec && tid < WBMt) aval = g.avec[m0 + tid];
#pragma unroll
        for (int x = 0; x < XT; ++x) {
            const int64_t row = m0 + 32 * XT * wh + 32 * x + li;
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    unsigned hi[4], lo[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {          // d = 0, 1: quad q; d = 2, 3: quad q + 1
                        const f32x2 v = {alpha * c[x][y][4 * q + 2 * d], alpha * c[x][y][4 * q + 2 * d + 1]};
                        const f16x2 fh = __builtin_convertvector(v, f16x2);
                        const f16x2 fl = __builtin_convertvector(v - __builtin_convertvector(fh, f32x2), f16x2);
                        hi[d] = __builtin_bit_cast(unsigned, fh); lo[d] = __builtin_bit_cast(unsigned, fl);
                    }
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hi[d]), "+v"(hi[2 + d]));
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo[d]), "+v"(lo[2 + d]));
                    }
                    const int64_t col = n0 + 64 * wq + 32 * y + 8 * (q + lk);
                    unsigned short* p = g.Cp + ((col >> 4) * g.M + row) * 16 + (col & 15);
                    const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]}, vl = {lo[0], lo[1], lo[2], lo[3]};
                    if (g.cp_nt & 1) { __builtin_nontemporal_store(vh, reinterpret_cast<u32x4*>(p)); __builtin_nontemporal_store(vl, reinterpret_cast<u32x4*>(p + g.pC)); }
                    else { *reinterpret_cast<u32x4*>(p) = vh; *reinterpret_cast<u32x4*>(p + g.pC) = vl; }
                }
        }
        if (g.Ct) {
            // The transposed planes: rows live on lanes, so a lane never holds two consecutive m of one n -- every 32 (m) x 64 (n) slice of
            // this wave goes through a wave-private LDS tile (the A ring is free between two work items): written row-wise (8 bytes = four n
            // per lane and quad), read back column-wise as 16-bit elements (all lanes of a read on one or two tile rows: conflict-free with
            // the 68-element row stride) into 16-byte units (n, eight consecutive m); a store instruction covers 32 n x 32 bytes = 1 KB.
            constexpr int RS = 68;                             // tile row stride in 16-bit elements (136 bytes)
            // (bare barriers: __syncthreads() would wait for vmcnt(0), i.e. for this item's plane stores to retire -- 14 us per item, measured)
            asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");      // every wave has left the k loop: the A slabs are dead
            unsigned short* wl = reinterpret_cast<unsigned short*>(&smem[0][0][0]) + wave * (32 * RS);
            if (g.avec) {
                if (tid < WBMt) alds[tid] = aval;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
            }
            const int nn = lane >> 1, hh = lane & 1;
            float ua[2] = {0.f, 0.f};
#pragma unroll
            for (int x = 0; x < XT; ++x) {
                const int64_t mrow = m0 + 32 * XT * wh + 32 * x;                    // first of this slice's 32 rows
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            unsigned w2[2];
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                const f32x2 v = {alpha * c[x][y][4 * q + 2 * d], alpha * c[x][y][4 * q + 2 * d + 1]};
                                const f16x2 fh = __builtin_convertvector(v, f16x2);
                                const f16x2 fo = pl == 0 ? fh : __builtin_convertvector(v - __builtin_convertvector(fh, f32x2), f16x2);
                                w2[d] = __builtin_bit_cast(unsigned, fo);
                            }
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            const u32x2 wv = {w2[0], w2[1]};
                            *reinterpret_cast<u32x2*>(wl + li * RS + 32 * y + 8 * q + 4 * lk) = wv;
                        }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            unsigned short e[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) e[j] = wl[(16 * mb + 8 * hh + j) * RS + 32 * nh + nn];
                            if (g.avec) {
                                const f32x4 a0 = *reinterpret_cast<const f32x4*>(alds + 32 * XT * wh + 32 * x + 16 * mb + 8 * hh);
                                const f32x4 a1 = *reinterpret_cast<const f32x4*>(alds + 32 * XT * wh + 32 * x + 16 * mb + 8 * hh + 4);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    ua[nh] = fmaf(a0[j], (float)__builtin_bit_cast(_Float16, e[j]), ua[nh]);
                                    ua[nh] = fmaf(a1[j], (float)__builtin_bit_cast(_Float16, e[4 + j]), ua[nh]);
                                }
                            }
                            const u32x4 ov = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                              (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
                            const int64_t n = n0 + 64 * wq + 32 * nh + nn;
                            unsigned short* p = g.Ct + (int64_t)pl * g.pCt + (((mrow >> 4) + mb) * g.N + n) * 16 + 8 * hh;
                            // (probe A/B at the bench shape: the LDS transposition costs 0.2 ms of the product, the 8.6 GB of stores 0.85 ms; non-temporal: no change)
                            if (g.cp_nt & 2) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<u32x4*>(p) = ov;
                        }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (g.avec) {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const float t2 = ua[nh] + __shfl_xor(ua[nh], 1, 64);            // the two 8-row halves of every 16-row block
                    if (hh == 0) g.Upart[((m0 + 32 * XT * wh) >> 7) * g.N + n0 + 64 * wq + 32 * nh + nn] = t2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");      // before the next work item's LDS-DMA lands in the ring
        }
    } else
#pragma unroll
    for (int x = 0; x < XT; ++x) {
        const int64_t row = m0 + 32 * XT * wh + 32 * x + li;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t col = n0 + 64 * wq + 32 * y + 8 * q + 4 * lk;
                float* p = g.c_blk ? g.C + ((col >> 4) * g.M + row) * 16 + (col & 15) : g.C + row * g.ldc + col;
                f32x4 v = {alpha * c[x][y][4 * q], alpha * c[x][y][4 * q + 1], alpha * c[x][y][4 * q + 2], alpha * c[x][y][4 * q + 3]};
                if (g.lower_only && col + 3 > row) {             // tile on the diagonal: element-wise
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e <= row) { if (atomic) atomic_add(p + e, v[e]); else p[e] = (beta == 0.f) ? v[e] : v[e] + beta * p[e]; }
                } else if (atomic) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomic_add(p + e, v[e]);
                } else if (beta == 0.f) {
                    *reinterpret_cast<f32x4*>(p) = v;      // (plain, not non-temporal: T of the SVGP step 12.7 -> 11.9 ms, same box)
                    if (g.maxout) cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                } else {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(p);
                    *reinterpret_cast<f32x4*>(p) = v + beta * o;
                }
            }
    }
    }   // work items
    if constexpr (FUSE) {
        if (fz_m0 >= 0) fz_flush_rows(fz_m0);
        float v = ((lane & 31) < g.fz.Q) ? fz_dl3 * fz_fl : 0.f;         // sum_n x_nq^2 C_n: lane (q = li) holds its share
        v += __shfl_xor(v, 32, 64);
        if (lane < 32 && lane < g.fz.Q) atomic_add(g.fz.dls3 + lane, (double)v);
    }
    if (g.maxout) {                                // one atomic per wave, and only if it can raise the word (non-negative floats order as their bits)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && __builtin_bit_cast(unsigned, cmax) > *(volatile unsigned*)g.maxout) atomicMax(g.maxout, __builtin_bit_cast(unsigned, cmax));
    }
}

// The kernels:  _128: 128 x 256 tiles, two workgroups per CU;  _256 (the default for 256-aligned shapes): 256 x 256, eight waves in two row
// halves;  _256w4: 256 x 256 by four 512-register waves;  _256pp: eight waves, ping-pong phases.  (The last two are measured alternatives
// kept for the probe build)
__global__ __launch_bounds__(256, 2) void gemm_f16x2_wide_kernel_128(SplitArgs g) { wide_body<4, 1, false>(g); }
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256(SplitArgs g) { wide_body<4, 2, false>(g); }
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256lo(SplitArgs g) { wide_body<4, 2, false, true, true>(g); }
#ifdef MXF_PROBES
// measured alternatives and killed experiments: in the PROBE library only (r06; until r05 they were compiled into the shipped one)
__global__ __launch_bounds__(256, 1) void gemm_f16x2_wide_kernel_256w4(SplitArgs g) { wide_body<8, 1, false>(g); }
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256pp(SplitArgs g) { wide_body<4, 2, true>(g); }
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256b1(SplitArgs g) { wide_body<4, 2, false, false>(g); }
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256bf(SplitArgs g) { wide_body<4, 2, false, true, false, false, true>(g); }
// the T product of the SVGP training call with the reverse pass as its epilogue (r05: correct, twice as slow -- 309 spilled registers)
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256fz(SplitArgs g) { wide_body<4, 2, false, true, false, false, false, true>(g); }
#endif
// planes-output forms (c_blk == 2; the whitened SVGP tier's V = L^-1 Kuf)
__global__ __launch_bounds__(512, 2) void gemm_f16x2_wide_kernel_256pl(SplitArgs g) { wide_body<4, 2, false, true, false, true>(g); }
__global__ __launch_bounds__(256, 2) void gemm_f16x2_wide_kernel_128pl(SplitArgs g) { wide_body<4, 1, false, true, false, true>(g); }

__global__ void split_scale_kernel(float* C, int64_t M, int64_t N, int64_t ldc, float beta, int lower_only) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col < N && !(lower_only && col > row)) {
        float* p = C + row * ldc + col;
        *p = (beta == 0.f) ? 0.f : beta * (*p);
    }
}

}  // namespace

size_t mxf_split_plane_elems(int64_t R, int64_t K) { return (size_t)((K + 15) / 16) * (size_t)R * 16; }

int mxf_maxabs_internal(mxf_ctx* h, int64_t R, int64_t K, const float* x, int64_t ld, unsigned* out, hipStream_t st, bool zero) {
    if (zero) MXF_HIP(h, hipMemsetAsync(out, 0, sizeof(unsigned), st));
    const int64_t n = R * K;
    if (n <= 0) return 0;
    int64_t nb = (n + 256 * 16 - 1) / (256 * 16);
    if (ld == K && n % 4 == 0 && ((uintptr_t)x % 16) == 0 && n >= (int64_t)1 << 22) {
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(maxabs_flat_kernel, dim3((unsigned)nb), dim3(256), 0, st, n / 4, reinterpret_cast<const uint4*>(x), out);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(maxabs_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, K, ld, x, out);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

int mxf_split_planes_internal(mxf_ctx* h, int64_t R, int64_t K, const float* X, int64_t ld, unsigned short* planes, hipStream_t st, int mode,
                              const unsigned* maxbits) {
    if (R <= 0 || K <= 0) return 0;
    const int64_t pstride = (int64_t)mxf_split_plane_elems(R, K);
    dim3 grid((unsigned)((K + 63) / 64), (unsigned)((R + 63) / 64));
    if (grid.y > 65535u) MXF_FAIL(h, -3, "split planes: too many rows for one launch (%lld)", (long long)R);
    if (mode == MXF_SPLIT_F16X2) hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, st, R, K, X, ld, planes, pstride, maxbits);
    else hipLaunchKernelGGL(split_planes_kernel<3>, grid, dim3(256), 0, st, R, K, X, ld, planes, pstride, (const unsigned*)nullptr);
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// C (M x N) = alpha * A (M x K) * B (N x K)^T + beta * C from split planes; pA / pB = plane strides in elements.  A k sub-range
// [k0, k0+K) of a (R x Ktot) operand is the pointer planes + (k0 / 16) * R * 16 with the FULL operand's plane stride.
int mxf_gemm_split_internal(mxf_ctx* h, int64_t M, int64_t N, int64_t K, double alpha, const unsigned short* A, int64_t pA,
                            const unsigned short* B, int64_t pB, double beta, float* C, int64_t ldc, int lower_only, hipStream_t st,
                            int reserve_cus, int mode, const float* ad0, int pow0, const unsigned* maxbits, const unsigned* maxbits2, int c_blocked,
                            unsigned* maxout, unsigned short* Cplanes, int64_t pC, int a_lower, unsigned short* Ct, int64_t pCt,
                            const float* avec, float* Upart, const mxf_fuse_args* fuse) {
    if (M <= 0 || N <= 0) return 0;
    SplitArgs g;
    memset(&g.fz, 0, sizeof(g.fz));
    if (fuse) {
        if (mode != MXF_SPLIT_F16X2 || (M % 256) != 0 || (N % WBN) != 0 || (fuse->B % 256) != 0 || Cplanes || lower_only || beta != 0.0 || K < 128)
            MXF_FAIL(h, -2, "mxf_gemm_split: the fused reverse pass needs the f16x2 format, M %% 256 == 0, whole 256-column tiles per sample and a plain product");
        g.fz = *fuse;
    }
    g.c_blk = c_blocked; g.maxout = nullptr;
    g.Cp = Cplanes; g.pC = pC; g.a_lower = a_lower; g.rot_div = 0;
    g.Ct = Ct; g.pCt = pCt; g.avec = avec; g.Upart = Upart;
    static const int cp_nt = (int)MXF_KNOB("MXF_SPLIT_CPNT", 0);
    g.cp_nt = cp_nt;
    if ((Ct && !Cplanes) || (avec && (!Ct || !Upart))) MXF_FAIL(h, -2, "mxf_gemm_split: the transposed planes come with the planes output, the partial sums with both");
    if (Cplanes) {
        if (mode != MXF_SPLIT_F16X2 || (M % 128) != 0 || (N % WBN) != 0 || beta != 0.0 || lower_only || c_blocked)
            MXF_FAIL(h, -2, "mxf_gemm_split: the planes output needs the f16x2 format, M %% 128 == 0, N %% 256 == 0, beta == 0 and a full product");
        g.c_blk = 2;
    } else if (a_lower) MXF_FAIL(h, -2, "mxf_gemm_split: a_lower is implemented for the planes output only");
    if (c_blocked && (N % 16 != 0 || beta != 0.0 || lower_only || ldc != N)) MXF_FAIL(h, -2, "mxf_gemm_split: the blocked output layout needs N %% 16 == 0, ldc == N, beta == 0 and a full product");
    g.ad0 = ad0; g.pow0 = pow0; g.maxbits = maxbits; g.maxbits2 = maxbits2;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K16 = (K + 15) / 16;
    g.pA = pA; g.pB = pB; g.ldc = ldc;
    g.alpha = (float)alpha; g.beta = (float)beta; g.lower_only = lower_only;
    static const int nprod = (int)MXF_KNOB("MXF_SPLIT_NPROD", 6);      // diagnostic: fewer products
    g.nprod = nprod;
    static const int use_dma = (int)MXF_KNOB("MXF_SPLIT_DMA", 1);      // 0: staged loads instead of LDS-DMA (128 x 128 kernel)
    g.use_dma = use_dma;
    static const int wide_env = (int)MXF_KNOB("MXF_SPLIT_WIDE", 3);
    // MXF_SPLIT_WIDE: 0 = never, 1 = whenever the shape allows, 2 = only the long-K lower-triangle products (Psi2), 3 (default) = those and
    // products written in 16-column blocks (T of the training step).  The kernel's epilogue puts ROWS on lanes: fine for a blocked C
    // (rows are 64 bytes apart) and for the small square Psi2, 16-byte pieces 4 N bytes apart for a wide row-major C -- the T shape then
    // takes 16.7 ms instead of 13.4 on the 128 x 128 kernel, blocked it takes 12.3.  (Before the kernel walked its work items persistently
    // the blocked T lost 1.9 ms on it as well.)
    const bool wide = Cplanes != nullptr || fuse != nullptr ||
                      (wide_env && (wide_env == 1 || lower_only || (wide_env == 3 && c_blocked)) && mode == MXF_SPLIT_F16X2 && g.use_dma && (M % 128) == 0 && (N % WBN) == 0 && (ldc % 4) == 0 &&
                       (((uintptr_t)C) % 16) == 0 && g.nprod >= 3 && (!lower_only || M == N));
    // rows per tile of the wide kernel: 256 when the shape allows, else 128 (four waves, two workgroups per CU).  MXF_SPLIT_XT: 4 = always
    // 128; 8 = 256 rows by four 512-register waves (one per SIMD); 16 (default) = 256 rows by eight waves, two row halves (two per SIMD)
    static const int xt_env = (int)MXF_KNOB("MXF_SPLIT_XT", 16);
#ifdef MXF_PROBES
    const int XT = (wide && !Cplanes && !fuse && xt_env == 8 && (M % 256) == 0) ? 8 : 4;
#else
    const int XT = 4;
#endif
    const int NH = (wide && (xt_env == 16 || Cplanes || fuse) && (M % 256) == 0) ? 2 : 1;
    const int64_t WBMh = 32 * XT * NH;
    int64_t tm = (M + SBM - 1) / SBM, tn = (N + SBN - 1) / SBN;
    if (lower_only && tm != tn) MXF_FAIL(h, -2, "mxf_gemm_split: lower_only needs a square output");
    int64_t tiles = lower_only ? tm * (tm + 1) / 2 : tm * tn;
    if (wide) {
        tm = M / WBMh; tn = N / WBN;
        tiles = 0;
        if (lower_only) { for (int64_t r = 0; r < tm; ++r) tiles += (WBMh * r + WBMh - 1) / WBN + 1; }
        else tiles = tm * tn;
    }
    int splitk = 1;
    // split-K target: ~one wave of workgroups (3 fit a CU; 3 or 4 per CU measure the same per step, 6 and more are slower; the wide kernel: 2,
    // its 256-row form 1).  A caller that reserves more than half of the chip wants a FEW workgroups next to other work -- phase A of Psi2,
    // sized for the four-per-CU kernel: ~216 workgroups, one per CU on 216 CUs, whichever kernel runs.
    const int64_t slots = wide ? (reserve_cus >= 128 ? (int64_t)(256 - reserve_cus) * 4 : (int64_t)(256 - reserve_cus) * (WBMh == 256 ? 1 : 2))
                               : (int64_t)(256 - reserve_cus) * (mode == MXF_SPLIT_F16X2 ? 4 : 3);
    if (tiles < slots && g.K16 >= 16 && !Cplanes && !fuse) {
        int64_t sk = slots / tiles;
        if (sk * tiles < (slots * 3) / 4) sk = (2 * slots) / tiles;
        const int64_t maxsplit = g.K16 / 8 > 0 ? g.K16 / 8 : 1;
        if (sk > maxsplit) sk = maxsplit;
        if (sk < 1) sk = 1;
        splitk = (int)sk;
    }
    int64_t kchunk = (g.K16 + splitk - 1) / splitk;
    if (kchunk < 1) kchunk = 1;
    splitk = (int)((g.K16 + kchunk - 1) / kchunk);
    if (splitk < 1) splitk = 1;
    g.splitk = splitk; g.kchunk = kchunk; g.atomic = splitk > 1;
    g.tm = tm; g.tn = tn; g.ntiles = tiles; g.nwg = tiles * splitk;
    // triangular planes-output products walk PAIRS of column strips (SplitArgs::pair); MXF_SPLIT_PAIR=0 (probe builds): the r04 rotation
    static const int pair_env = (int)MXF_KNOB("MXF_SPLIT_PAIR", 1);
    g.pair = (wide && Cplanes && a_lower && pair_env && tm >= 2 && tn >= 2) ? 1 : 0;
    if (g.pair) { g.ntiles = tm * ((tn + 1) / 2); g.nwg = g.ntiles; }
    g.sync = nullptr; g.sync_n = 1; g.sync_period = 0; g.sync_slots = 0;
    if (g.nwg > 2147483647LL) MXF_FAIL(h, -3, "mxf_gemm_split: grid too large");
    if (g.atomic && beta != 1.0) {
        if (M > 65535) MXF_FAIL(h, -3, "mxf_gemm_split: split-K path needs M<=65535");
        dim3 gs((unsigned)((N + 255) / 256), (unsigned)M);
        hipLaunchKernelGGL(split_scale_kernel, gs, dim3(256), 0, st, C, M, N, ldc, (float)beta, lower_only);
    }
    const bool dma = g.use_dma && (M % SBM) == 0 && (N % SBN) == 0;
    if (wide) {
        g.maxout = (splitk == 1 && beta == 0.0 && !lower_only) ? maxout : nullptr;       // (other paths leave the word as it is: the caller sees 0)
        // persistent: as many workgroups as fit the chip (the kernel's occupancy) walk the work items; fewer items than that: one each
        static const int64_t wide_grid_env = MXF_KNOB("MXF_SPLIT_WIDE_GRID", 0);
        // (planes-output products honour reserve_cus: a caller that runs a latency-bound chain next to this product leaves it some CUs)
        int64_t wide_grid = wide_grid_env > 0 ? wide_grid_env : (WBMh == 256 ? (Cplanes ? (256 - reserve_cus) / 8 * 8 : 256) : (Cplanes ? (256 - reserve_cus) / 8 * 8 * 2 : 512));
        // paired items: the tm roles of a pair must be resident in the same persistent round -- workgroups per XCD a multiple of tm
        if (g.pair && wide_grid / 8 >= tm) wide_grid = (wide_grid / 8) / tm * tm * 8;
        const int64_t grid = (wide_grid >= 8 && g.nwg > wide_grid) ? wide_grid / 8 * 8 : g.nwg;
        // rendezvous groups (wg_rendezvous): MXF_SPLIT_SYNC 0 = none; 1 = the row tiles of one column strip (full products) / the tiles of
        // one k split (split-K products); 2 = full products: all workgroups of an XCD, once per work item
        static const int sync_env = (int)MXF_KNOB("MXF_SPLIT_SYNC", 1);
        static const int sync_period_env = (int)MXF_KNOB("MXF_SPLIT_SYNC_PERIOD", 16);
        if (g.pair) {
            // the tm roles of a strip pair = tm consecutive items of one XCD's run, taken in the same round by tm different workgroups: they
            // start together (bounded rendezvous, a pacing hint) and then stay in step by construction -- every role does tm + 1 units
            // (r05 measurement: the pairing alone takes the fabric fetch from 21.5 to 13.5 GB -- with or without the rendezvous -- and the
            //  product is 0.1 ms faster without it: 8.40 vs 8.50 ms.  Off by default; MXF_SPLIT_PAIR_SYNC=1 in probe builds.)
            static const int pair_sync_env = (int)MXF_KNOB("MXF_SPLIT_PAIR_SYNC", 0);
            const int64_t q = g.nwg / 8, per_xcd = grid / 8;
            if (sync_env && pair_sync_env && g.nwg % 8 == 0 && q % tm == 0 && per_xcd >= tm && per_xcd % tm == 0) {
                g.sync = mxf_gsync(h, (unsigned)(g.nwg / tm));
                if (g.sync) { g.sync_n = (int)tm; g.sync_period = 0; g.sync_slots = 0; }
            }
        } else if (Cplanes && a_lower) {
            g.rot_div = (grid / 8) / tm > 0 ? (grid / 8) / tm : 1;       // (no rendezvous: the row tiles of a strip carry unequal work)
        } else if (sync_env && !lower_only && splitk == 1 && g.nwg % 8 == 0 && g.nwg >= 16) {
            const int64_t q = g.nwg / 8, per_xcd = grid / 8;        // work items / resident workgroups per XCD
            int64_t n = sync_env == 2 ? per_xcd : tm;
            // a group = n consecutive work items of one XCD's run, taken in the same persistent round by n different workgroups
            const bool ok = n >= 2 && q % n == 0 && per_xcd % n == 0 && (g.nwg <= grid || g.nwg % grid == 0);
            if (ok) {
                g.sync = mxf_gsync(h, (unsigned)(g.nwg / n));
                g.sync_n = (int)n; g.sync_period = 0; g.sync_slots = 0;
            }
        } else if (sync_env && splitk > 1 && g.nwg <= grid && tiles >= 2 && sync_period_env > 0) {
            // every tile of a k split is resident at once: they meet every `period` trips of three k blocks
            const int64_t trips = kchunk / 3;
            int64_t period = sync_period_env;
            int64_t slots_per = trips / period;
            while (slots_per * splitk > (int64_t)(MXF_NGSYNC / 8)) { period *= 2; slots_per = trips / period; }
            if (slots_per >= 1) {
                g.sync = mxf_gsync(h, (unsigned)(slots_per * splitk));
                g.sync_n = (int)tiles; g.sync_period = (int)period; g.sync_slots = (int)slots_per;
            }
        }
        static const int pp_env = (int)MXF_KNOB("MXF_SPLIT_PP", 0);        // ping-pong phases of the two row halves (NH = 2)
        static const int lskip_env = (int)MXF_KNOB("MXF_SPLIT_LSKIP", 1);  // lower-only products: the waves above the diagonal idle (see wide_body)
        static const int bhi_env = (int)MXF_KNOB("MXF_SPLIT_BHI", 0);      // experiment: B through its high plane only, blocked-output products
#ifdef MXF_PROBES
        static const int bfi_env = (int)MXF_KNOB("MXF_SPLIT_BF16MFMA", 0);
        if (bfi_env && NH == 2 && !Cplanes && !lower_only) { hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256bf, dim3((unsigned)grid), dim3(512), 0, st, g); MXF_LAUNCH_CHECK(h); return 0; }
#endif
#ifdef MXF_PROBES
        if (fuse) { hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256fz, dim3((unsigned)grid), dim3(512), 0, st, g); MXF_LAUNCH_CHECK(h); return 0; }
        if (!Cplanes && NH == 2 && bhi_env && c_blocked) { hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256b1, dim3((unsigned)grid), dim3(512), 0, st, g); MXF_LAUNCH_CHECK(h); return 0; }
        if (!Cplanes && NH == 2 && pp_env) { hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256pp, dim3((unsigned)grid), dim3(512), 0, st, g); MXF_LAUNCH_CHECK(h); return 0; }
        if (!Cplanes && NH != 2 && XT == 8) { hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256w4, dim3((unsigned)grid), dim3(256), 0, st, g); MXF_LAUNCH_CHECK(h); return 0; }
#else
        if (fuse) MXF_FAIL(h, -3, "mxf_gemm_split: the fused reverse pass exists in the probe build only");
        (void)bhi_env; (void)pp_env;
#endif
        if (Cplanes && NH == 2) hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256pl, dim3((unsigned)grid), dim3(512), 0, st, g);
        else if (Cplanes) hipLaunchKernelGGL(gemm_f16x2_wide_kernel_128pl, dim3((unsigned)grid), dim3(256), 0, st, g);
        else if (NH == 2 && lower_only && lskip_env) hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256lo, dim3((unsigned)grid), dim3(512), 0, st, g);
        else if (NH == 2) hipLaunchKernelGGL(gemm_f16x2_wide_kernel_256, dim3((unsigned)grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL(gemm_f16x2_wide_kernel_128, dim3((unsigned)grid), dim3(256), 0, st, g);
        MXF_LAUNCH_CHECK(h);
        return 0;
    }
    if (mode == MXF_SPLIT_F16X2) {
        if (dma) hipLaunchKernelGGL((gemm_split_kernel<true, 2>), dim3((unsigned)g.nwg), dim3(SNT), 0, st, g);
        else hipLaunchKernelGGL((gemm_split_kernel<false, 2>), dim3((unsigned)g.nwg), dim3(SNT), 0, st, g);
    } else {
        if (dma) hipLaunchKernelGGL((gemm_split_kernel<true, 3>), dim3((unsigned)g.nwg), dim3(SNT), 0, st, g);
        else hipLaunchKernelGGL((gemm_split_kernel<false, 3>), dim3((unsigned)g.nwg), dim3(SNT), 0, st, g);
    }
    MXF_LAUNCH_CHECK(h);
    return 0;
}

// C ABI (f32 only): operands given as plain f32 matrices; they are split into the handle's scratch and multiplied.
extern "C" int mxf_gemm_f32x3(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda, const void* B,
                              int64_t ldb, double beta, void* C, int64_t ldc, int lower_only, void* stream) {
    if (!h) return -1;
    if (M <= 0 || N <= 0 || K <= 0) MXF_FAIL(h, -2, "mxf_gemm_f32x3: bad shape");
    if (!A || !B || !C) MXF_FAIL(h, -2, "mxf_gemm_f32x3: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t ea = mxf_split_plane_elems(M, K), eb = mxf_split_plane_elems(N, K);
    const size_t need = mxf_align(3 * ea * 2) + mxf_align(3 * eb * 2);
    char* ws = (char*)mxf_ws(h, need);
    if (!ws) MXF_FAIL(h, -4, "mxf_gemm_f32x3: cannot allocate %zu bytes of scratch", need);
    unsigned short* pa = (unsigned short*)ws;
    unsigned short* pb = (unsigned short*)(ws + mxf_align(3 * ea * 2));
    int rc = mxf_split_planes_internal(h, M, K, (const float*)A, lda, pa, st);
    if (rc) return rc;
    rc = mxf_split_planes_internal(h, N, K, (const float*)B, ldb, pb, st);
    if (rc) return rc;
    return mxf_gemm_split_internal(h, M, N, K, alpha, pa, (int64_t)ea, pb, (int64_t)eb, beta, (float*)C, ldc, lower_only, st, 0);
}

// The same product from two scaled f16 terms per operand and three MFMA products (see the header of this file): each operand is scaled
// by the power of two that puts its largest magnitude at [2^13, 2^14).  Normwise f32 accuracy; an element more than 2^18 below its
// operand's maximum keeps fewer than 22 bits (absolute error <= 2^-39 of the maximum).
extern "C" int mxf_gemm_f16x2(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A, int64_t lda, const void* B,
                              int64_t ldb, double beta, void* C, int64_t ldc, int lower_only, void* stream) {
    if (!h) return -1;
    if (M <= 0 || N <= 0 || K <= 0) MXF_FAIL(h, -2, "mxf_gemm_f16x2: bad shape");
    if (!A || !B || !C) MXF_FAIL(h, -2, "mxf_gemm_f16x2: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t ea = mxf_split_plane_elems(M, K), eb = mxf_split_plane_elems(N, K);
    const size_t need = mxf_align(2 * ea * 2) + mxf_align(2 * eb * 2) + mxf_align(2 * sizeof(unsigned));
    char* ws = (char*)mxf_ws(h, need);
    if (!ws) MXF_FAIL(h, -4, "mxf_gemm_f16x2: cannot allocate %zu bytes of scratch", need);
    unsigned short* pa = (unsigned short*)ws;
    unsigned short* pb = (unsigned short*)(ws + mxf_align(2 * ea * 2));
    unsigned* mx = (unsigned*)(ws + mxf_align(2 * ea * 2) + mxf_align(2 * eb * 2));
    int rc = mxf_maxabs_internal(h, M, K, (const float*)A, lda, mx, st);
    if (rc) return rc;
    rc = mxf_maxabs_internal(h, N, K, (const float*)B, ldb, mx + 1, st);
    if (rc) return rc;
    rc = mxf_split_planes_internal(h, M, K, (const float*)A, lda, pa, st, MXF_SPLIT_F16X2, mx);
    if (rc) return rc;
    rc = mxf_split_planes_internal(h, N, K, (const float*)B, ldb, pb, st, MXF_SPLIT_F16X2, mx + 1);
    if (rc) return rc;
    return mxf_gemm_split_internal(h, M, N, K, alpha, pa, (int64_t)ea, pb, (int64_t)eb, beta, (float*)C, ldc, lower_only, st, 0, MXF_SPLIT_F16X2,
                                   nullptr, 0, mx, mx + 1);
}

// the two halves of mxf_gemm_f16x2 for callers that reuse split operands: planes = 2 * mxf_f32x3_plane_elems(R, K) 16-bit elements,
// maxword = one 32-bit device word (bit pattern of max |X|, from which the operand's power-of-two scale is derived)
extern "C" int mxf_f16x2_split(mxf_handle h, int64_t R, int64_t K, const void* X, int64_t ld, void* planes, void* maxword, void* stream) {
    if (!h) return -1;
    if (R <= 0 || K <= 0 || !X || !planes || !maxword) MXF_FAIL(h, -2, "mxf_f16x2_split: bad argument");
    int rc = mxf_maxabs_internal(h, R, K, (const float*)X, ld, (unsigned*)maxword, (hipStream_t)stream);
    if (rc) return rc;
    return mxf_split_planes_internal(h, R, K, (const float*)X, ld, (unsigned short*)planes, (hipStream_t)stream, MXF_SPLIT_F16X2, (const unsigned*)maxword);
}

extern "C" int mxf_gemm_f16x2_planes(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* A_maxword,
                                     const void* B_planes, const void* B_maxword, double beta, void* C, int64_t ldc, int lower_only, void* stream) {
    if (!h) return -1;
    if (M <= 0 || N <= 0 || K <= 0 || !A_planes || !B_planes || !A_maxword || !B_maxword || !C) MXF_FAIL(h, -2, "mxf_gemm_f16x2_planes: bad argument");
    const int blocked = lower_only == 2;            // lower_only = 2: full product, C in 16-column blocks (include/mxf_gp.h)
    return mxf_gemm_split_internal(h, M, N, K, alpha, (const unsigned short*)A_planes, (int64_t)mxf_split_plane_elems(M, K),
                                   (const unsigned short*)B_planes, (int64_t)mxf_split_plane_elems(N, K), beta, (float*)C, blocked ? N : ldc,
                                   blocked ? 0 : lower_only, (hipStream_t)stream, 0, MXF_SPLIT_F16X2, nullptr, 0, (const unsigned*)A_maxword,
                                   (const unsigned*)B_maxword, blocked);
}

// the two halves of mxf_gemm_f32x3 for callers that reuse split operands: planes = 3 * mxf_f32x3_plane_elems(R, K) bf16 (uint16) elements
extern "C" int64_t mxf_f32x3_plane_elems(int64_t R, int64_t K) { return (R > 0 && K > 0) ? (int64_t)mxf_split_plane_elems(R, K) : 0; }

extern "C" int mxf_f32x3_split(mxf_handle h, int64_t R, int64_t K, const void* X, int64_t ld, void* planes, void* stream) {
    if (!h) return -1;
    if (R <= 0 || K <= 0 || !X || !planes) MXF_FAIL(h, -2, "mxf_f32x3_split: bad argument");
    return mxf_split_planes_internal(h, R, K, (const float*)X, ld, (unsigned short*)planes, (hipStream_t)stream);
}

extern "C" int mxf_gemm_f32x3_planes(mxf_handle h, int64_t M, int64_t N, int64_t K, double alpha, const void* A_planes, const void* B_planes,
                                     double beta, void* C, int64_t ldc, int lower_only, void* stream) {
    if (!h) return -1;
    if (M <= 0 || N <= 0 || K <= 0 || !A_planes || !B_planes || !C) MXF_FAIL(h, -2, "mxf_gemm_f32x3_planes: bad argument");
    return mxf_gemm_split_internal(h, M, N, K, alpha, (const unsigned short*)A_planes, (int64_t)mxf_split_plane_elems(M, K),
                                   (const unsigned short*)B_planes, (int64_t)mxf_split_plane_elems(N, K), beta, (float*)C, ldc, lower_only,
                                   (hipStream_t)stream, 0);
}

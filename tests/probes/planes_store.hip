// r03 probe: the store pattern of the split-plane Gram passes WITHOUT the covariance arithmetic (store-only twins) -- what does the write
// path alone sustain for 8.6 GB in 16-byte pieces?   hipcc --offload-arch=gfx950 -O3 planes_store.hip -o /tmp/planes_store && /tmp/planes_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: nontemporal, two planes; 1: plain stores, two planes; 2: nontemporal, ONE plane of twice the piece (32 B per row contiguous hi|lo)
// 3: two planes, nontemporal, but each wave writes 4 KB contiguous per plane (64 lanes x 4 consecutive instructions on consecutive 1 KB)
template <int MODE>
__global__ __launch_bounds__(64) void store_kernel(int64_t R, int64_t K16, unsigned short* __restrict__ P, int64_t pstride, int kbpb, unsigned seed) {
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int64_t ra = r0 + (lane & 31), rb = ra + 32;
    const int hsel = lane >> 5;
    const int64_t kb0 = (int64_t)blockIdx.y * kbpb;
    u32x4 v = {seed + lane, seed * 3u, seed ^ (unsigned)blockIdx.x, seed + (unsigned)blockIdx.y};
    for (int64_t kb = kb0; kb < kb0 + kbpb && kb < K16; ++kb) {
        v.x += 1u;
        if (MODE == 2) {
            unsigned short* base = P + (kb * R) * 32;
            u32x4* d = reinterpret_cast<u32x4*>(base + (r0 * 32)) + lane;          // 64 lanes x 16 B = 1 KB, four of them = 64 rows x 64 B
            __builtin_nontemporal_store(v, d);
            __builtin_nontemporal_store(v, d + 64);
            __builtin_nontemporal_store(v, d + 128);
            __builtin_nontemporal_store(v, d + 192);
        } else {
            unsigned short* base = P + (kb * R) * 16 + hsel * 8;
            u32x4* a0 = reinterpret_cast<u32x4*>(base + ra * 16);
            u32x4* a1 = reinterpret_cast<u32x4*>(base + ra * 16 + pstride);
            u32x4* b0 = reinterpret_cast<u32x4*>(base + rb * 16);
            u32x4* b1 = reinterpret_cast<u32x4*>(base + rb * 16 + pstride);
            if (MODE == 0) {
                __builtin_nontemporal_store(v, a0); __builtin_nontemporal_store(v, a1);
                __builtin_nontemporal_store(v, b0); __builtin_nontemporal_store(v, b1);
            } else { *a0 = v; *a1 = v; *b0 = v; *b1 = v; }
        }
    }
}
// 256-thread blocks, thread <-> (row, half) as gram_planes_kernel, 128 rows per block
template <int NT>
__global__ __launch_bounds__(256) void store256_kernel(int64_t R, int64_t K16, unsigned short* __restrict__ P, int64_t pstride, int kbpb, unsigned seed) {
    const int tid = threadIdx.x, rl = tid >> 1, half = tid & 1;
    const int64_t r = (int64_t)blockIdx.x * 128 + rl;
    const int64_t kb0 = (int64_t)blockIdx.y * kbpb;
    u32x4 v = {seed + tid, seed * 3u, seed ^ (unsigned)blockIdx.x, seed + (unsigned)blockIdx.y};
    for (int64_t kb = kb0; kb < kb0 + kbpb && kb < K16; ++kb) {
        v.x += 1u;
        unsigned short* dst = P + (kb * R + r) * 16 + half * 8;
        if (NT) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst)); __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + pstride)); }
        else { *reinterpret_cast<u32x4*>(dst) = v; *reinterpret_cast<u32x4*>(dst + pstride) = v; }
    }
}
int main() {
    const int64_t R = 1024, K = 2097152, K16 = K / 16;
    const int64_t pstride = R * K;            // elements (2 B)
    unsigned short* P; hipMalloc(&P, (size_t)pstride * 2 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int n = 5;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
        printf("%-48s %.3f ms  %.2f TB/s\n", name, ms, (double)pstride * 4 / ms * 1e-9);
    };
    for (int kbpb : {4, 8, 32}) {
        dim3 g((unsigned)(R / 64), (unsigned)(K16 / kbpb));
        char nm[96];
        snprintf(nm, 96, "one-wave nt 2 planes, R=1024, kb/blk=%d", kbpb);   timeit(nm, [&] { hipLaunchKernelGGL(store_kernel<0>, g, dim3(64), 0, 0, R, K16, P, pstride, kbpb, 1u); });
        snprintf(nm, 96, "one-wave plain 2 planes, R=1024, kb/blk=%d", kbpb); timeit(nm, [&] { hipLaunchKernelGGL(store_kernel<1>, g, dim3(64), 0, 0, R, K16, P, pstride, kbpb, 1u); });
        snprintf(nm, 96, "one-wave nt 1 plane (hi|lo), R=1024, kb/blk=%d", kbpb); timeit(nm, [&] { hipLaunchKernelGGL(store_kernel<2>, g, dim3(64), 0, 0, R, K16, P, pstride, kbpb, 1u); });
    }
    {   // the other orientation: R = 2.1 M rows, 64 k blocks, each block walks all of them
        const int64_t R2 = K, K162 = 64;
        dim3 g((unsigned)(R2 / 64), 1);
        timeit("one-wave nt 2 planes, R=2.1M, 64 kb", [&] { hipLaunchKernelGGL(store_kernel<0>, g, dim3(64), 0, 0, R2, K162, P, pstride, 64, 1u); });
        timeit("one-wave plain 2 planes, R=2.1M, 64 kb", [&] { hipLaunchKernelGGL(store_kernel<1>, g, dim3(64), 0, 0, R2, K162, P, pstride, 64, 1u); });
        dim3 g4((unsigned)(R2 / 64), 4);
        timeit("one-wave nt 2 planes, R=2.1M, 16 kb x4", [&] { hipLaunchKernelGGL(store_kernel<0>, g4, dim3(64), 0, 0, R2, K162, P, pstride, 16, 1u); });
        dim3 g2((unsigned)(R2 / 128), 1);
        timeit("256-thread nt 2 planes, R=2.1M", [&] { hipLaunchKernelGGL(store256_kernel<1>, g2, dim3(256), 0, 0, R2, K162, P, pstride, 64, 1u); });
    }
    {
        dim3 g((unsigned)(R / 128), (unsigned)(K16 / 16));
        timeit("256-thread nt 2 planes, R=1024, 16 kb/blk", [&] { hipLaunchKernelGGL(store256_kernel<1>, g, dim3(256), 0, 0, R, K16, P, pstride, 16, 1u); });
        timeit("256-thread plain 2 planes, R=1024, 16 kb/blk", [&] { hipLaunchKernelGGL(store256_kernel<0>, g, dim3(256), 0, 0, R, K16, P, pstride, 16, 1u); });
    }
    hipMemset(P, 0, 16); 
    timeit("hipMemsetAsync 8.6 GB", [&] { hipMemsetAsync(P, 1, (size_t)pstride * 4, 0); });
    return 0;
}

// HBM-write ceiling micro-benchmark for the Gram kernel's store patterns (MI355X).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tile pattern: block (4 waves) covers TR rows x (4*64*4*CW) cols; each lane stores CW x 16 B per row
template <int TR, int CW, bool NT>
__global__ __launch_bounds__(256) void fill_tile(float* __restrict__ K, int64_t N, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * (256 * CW) + lane * 4;
    f32x4 o = {v, v + 1, v + 2, v + 3};
    for (int r = 0; r < TR; ++r) {
        float* dst = K + (row0 + r) * N + col0;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst + c * 256));
            else *reinterpret_cast<f32x4*>(dst + c * 256) = o;
        }
        o.x += 1.f;
    }
}
// linear: grid-stride fully linear fill
template <bool NT>
__global__ __launch_bounds__(256) void fill_linear(float* __restrict__ K, int64_t n4, float v) {
    f32x4 o = {v, v, v, v};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(K) + i);
        else reinterpret_cast<f32x4*>(K)[i] = o;
    }
}
template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int64_t N = 65536; float* K; hipMalloc(&K, N * N * 4);
    const double gb = (double)N * N * 4 / 1e9;
#define RUN(name, ...) { float ms = timeit([&] { __VA_ARGS__; }); printf("%-28s %8.3f ms %8.1f GB/s\n", name, ms, gb / ms * 1e3); }
    RUN("tile TR=64 CW=1 nt", (fill_tile<64, 1, true><<<dim3(N / 1024, N / 64), 256>>>(K, N, 1.f)));
    RUN("tile TR=64 CW=1 plain", (fill_tile<64, 1, false><<<dim3(N / 1024, N / 64), 256>>>(K, N, 1.f)));
    RUN("tile TR=16 CW=1 nt", (fill_tile<16, 1, true><<<dim3(N / 1024, N / 16), 256>>>(K, N, 1.f)));
    RUN("tile TR=256 CW=1 nt", (fill_tile<256, 1, true><<<dim3(N / 1024, N / 256), 256>>>(K, N, 1.f)));
    RUN("tile TR=64 CW=2 nt", (fill_tile<64, 2, true><<<dim3(N / 2048, N / 64), 256>>>(K, N, 1.f)));
    RUN("tile TR=64 CW=4 nt", (fill_tile<64, 4, true><<<dim3(N / 4096, N / 64), 256>>>(K, N, 1.f)));
    RUN("tile TR=32 CW=4 nt", (fill_tile<32, 4, true><<<dim3(N / 4096, N / 32), 256>>>(K, N, 1.f)));
    RUN("tile TR=64 CW=4 plain", (fill_tile<64, 4, false><<<dim3(N / 4096, N / 64), 256>>>(K, N, 1.f)));
    RUN("tile TR=8 CW=16 nt", (fill_tile<8, 16, true><<<dim3(N / 16384, N / 8), 256>>>(K, N, 1.f)));
    RUN("linear nt 2048 blocks", (fill_linear<true><<<2048, 256>>>(K, N * N / 4, 1.f)));
    RUN("linear plain 2048 blocks", (fill_linear<false><<<2048, 256>>>(K, N * N / 4, 1.f)));
    RUN("linear nt 16384 blocks", (fill_linear<true><<<16384, 256>>>(K, N * N / 4, 1.f)));
    hipMemset(K, 0, 1 << 20);
    RUN("hipMemsetAsync", hipMemsetAsync(K, 0, N * N * 4, 0));
    return 0;
}

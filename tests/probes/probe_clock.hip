// Shader clock under a light load: one wave runs a chain of dependent v_fma_f32 (4 cycles each on a SIMD16 at wave64) and times it with
// the 100 MHz wall clock; then the same with all CUs busy (a spinning kernel on another stream).  hipcc --offload-arch=gfx950 probe_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, long long* t, int n) {
    float x = out[0];
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void burn(float* out, int n) {
    float x = out[threadIdx.x], y = x + 1.f, z = x + 2.f, w = x + 3.f;
    for (int i = 0; i < n; ++i) { x = __builtin_fmaf(x, 1.0000001f, 1e-9f); y = __builtin_fmaf(y, 1.0000001f, 1e-9f); z = __builtin_fmaf(z, 1.0000001f, 1e-9f); w = __builtin_fmaf(w, 1.0000001f, 1e-9f); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}
int main() {
    float* d; long long* t; hipMalloc(&d, 1 << 24); hipMalloc(&t, 64); hipMemset(d, 0, 1 << 24);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int rep = 0; rep < 4; ++rep) {
        const int n = 20000 << rep;            // 1.28M .. 10M dependent FMAs
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, n);
        hipStreamSynchronize(s1);
        long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("alone : %8d x 64 dependent FMAs in %8.1f us -> %.0f MHz (at 4 cycles each)\n", n, h / 100.0, 4.0 * 64 * n / (h / 100.0));
    }
    hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, s2, d + 65536, 40000000);
    for (int rep = 0; rep < 3; ++rep) {
        const int n = 40000;
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, n);
        hipStreamSynchronize(s1);
        long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("loaded: %8d x 64 dependent FMAs in %8.1f us -> %.0f MHz\n", n, h / 100.0, 4.0 * 64 * n / (h / 100.0));
    }
    hipDeviceSynchronize();
    return 0;
}

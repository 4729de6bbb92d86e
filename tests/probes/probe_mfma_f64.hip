// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: every wave runs independent accumulator chains, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 probe_mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    f64x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f64x4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it += 32) {      // (32 rounds per trip: the compiler shuffles the accumulators between AGPRs and VGPRs at the loop edge)
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs, int threads, const char* name) {
    double* d; hipMalloc(&d, sizeof(double) * wgs * threads);
    const int iters = 20000 / 32 * 32;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, d, 128);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * (threads / 64) * iters * NACC * 2048.0;
    printf("%-40s %8.2f TFLOP/s  (%.0f cycles per MFMA per SIMD at 2.4 GHz)\n", name, flops / ms / 1e9,
           2.4e9 * (ms * 1e-3) / ((double)wgs * (threads / 64) / 1024.0 * iters * NACC));
    hipFree(d);
}
int main() {
    run<8>(256, 256, "1 wave/SIMD, 8 accumulators");
    run<8>(512, 256, "2 waves/SIMD, 8 accumulators");
    run<4>(512, 256, "2 waves/SIMD, 4 accumulators");
    run<8>(1024, 256, "4 waves/SIMD, 8 accumulators");
    return 0;
}

// probe: cross-row (16-lane) adds with the gfx950 v_permlane16_swap / v_permlane32_swap instructions (inline asm; the clang builtin
// folds the two results into one register on ROCm 7.2 and returns 2*x).  Expected output: every lane 1111 + 0.004*(lane%16).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float xrow_add16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xrow_add32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(float* o) {
    float v = o[threadIdx.x];
    o[threadIdx.x] = xrow_add32(xrow_add16(v));
}
int main() {
    float h[64], *d;
    for (int i = 0; i < 64; ++i) h[i] = (i / 16 == 0 ? 1 : i / 16 == 1 ? 10 : i / 16 == 2 ? 100 : 1000) + 0.001f * (i % 16);
    if (hipMalloc(&d, 256) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 5) printf("%d:%g ", i, h[i]);
    printf("\n");
    return 0;
}

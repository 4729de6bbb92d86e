// r03 probe: (1) does a matrix-pipe instruction of one wave run UNDER the VALU work of another wave of the same SIMD?  f32 MFMA
// (v_mfma_f32_16x16x4_f32) vs f16 MFMA (v_mfma_f32_16x16x16_f16);  (2) does the f16 MFMA keep f16 SUBNORMAL inputs?
// hipcc --offload-arch=gfx950 -O3 probe_mfma_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: even waves run VALU chains; bit 1: odd waves run MFMAs (KIND 0 = f32 16x16x4, 1 = f16 16x16x16)
template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    float acc = threadIdx.x * 1e-3f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f}, c2 = c, c3 = c, c4 = c;
    if ((wave & 1) == 0) {
        if (mode & 1) {
            float a0 = acc, a1 = acc + 1.f, a2 = acc + 2.f, a3 = acc + 3.f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) { a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f); }
            }
            acc = a0 + a1 + a2 + a3;
        }
    } else if (mode & 2) {
        if (KIND == 0) {
            const float a = acc, b = acc * 0.5f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
                }
            }
        } else {
            const f16x4 a = {(_Float16)acc, (_Float16)1.f, (_Float16)2.f, (_Float16)0.5f}, b = {(_Float16)0.25f, (_Float16)acc, (_Float16)1.f, (_Float16)2.f};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c3, 0, 0, 0); c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c4, 0, 0, 0);
                }
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc + c[0] + c2[1] + c3[2] + c4[3];
}

__global__ void denorm(float* out) {
    const int l = threadIdx.x;
    // A[i][k] = 2^-20 (f16 subnormal), B[k][j] = 1024: D[i][j] = 16 * 2^-10 if subnormals are kept, 0 if flushed
    const _Float16 tiny = (_Float16)9.5367431640625e-07f;
    const f16x4 a = {tiny, tiny, tiny, tiny}, b = {(_Float16)1024.f, (_Float16)1024.f, (_Float16)1024.f, (_Float16)1024.f};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    out[l] = c[0];
}

int main() {
    float* out; hipMalloc(&out, 2048 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int kind = 0; kind < 2; ++kind) {
        float t[4] = {0, 0, 0, 0};
        for (int mode = 1; mode <= 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, mode, iters, out);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, mode, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&t[mode], e0, e1);
            }
        }
        printf("%s MFMA: VALU waves alone %.3f ms, MFMA waves alone %.3f ms, both at once %.3f ms (sum %.3f)\n", kind ? "f16 16x16x16" : "f32 16x16x4 ", t[1], t[2], t[3], t[1] + t[2]);
    }
    hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, out);
    float h[64]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
    printf("f16 subnormal inputs: D = %g (kept: %g, flushed: 0)\n", h[0], 16 * 9.5367431640625e-07 * 1024);
    return 0;
}

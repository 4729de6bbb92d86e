// r04 A/B: cache-policy bits of the Gram kernel's 16-byte stores (gfx950 global_store sc0 / sc1 / nt) on the production mapping -- one-wave
// workgroups, 12 rows each, lane <-> 4 columns, x rows through scalar loads -- at N = 65536, Q = 8, f32 RBF.  One process, interleaved rounds.
// Build: hipcc --offload-arch=gfx950 -O3 -o gram_store_policy gram_store_policy.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int QT = 8;

__device__ __forceinline__ f32x4 rbf_row(const float (&x)[QT], const float (&z)[4][QT]) {
    f32x4 out;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 acc2 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < QT; ++q) { const f32x2 xx = {x[q], x[q]}, zz = {z[2 * p][q], z[2 * p + 1][q]}; const f32x2 d = xx - zz; acc2 = __builtin_elementwise_fma(d, d, acc2); }
        out[2 * p] = __builtin_amdgcn_exp2f(-acc2.x);
        out[2 * p + 1] = __builtin_amdgcn_exp2f(-acc2.y);
    }
    return out;
}

// POL: 0 plain, 1 __builtin_nontemporal_store (nt), 2 "sc0 sc1", 3 "sc1 nt", 4 "sc0 sc1 nt", 5 "sc1", 6 "sc0 nt"
template <int TR, int POL>
__global__ __launch_bounds__(64) void gram_pol(const float* __restrict__ Xs, float* __restrict__ K, int64_t N) {
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.y * TR, col0 = (int64_t)blockIdx.x * 256 + lane * 4;
    float z[4][QT];
#pragma unroll
    for (int i = 0; i < 4 * QT; i += 4) *reinterpret_cast<f32x4*>(&z[0][0] + i) = *reinterpret_cast<const f32x4*>(Xs + col0 * QT + i);
#pragma unroll 2
    for (int r = 0; r < TR; ++r) {
        float x[QT];
        const float* Xr = Xs + (row0 + r) * QT;
#pragma unroll
        for (int q = 0; q < QT; ++q) x[q] = Xr[q];
        const f32x4 out = rbf_row(x, z);
        float* dst = K + (row0 + r) * N + col0;
        if (POL == 0) *reinterpret_cast<f32x4*>(dst) = out;
        else if (POL == 1) __builtin_nontemporal_store(out, reinterpret_cast<f32x4*>(dst));
        else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(out) : "memory");
        else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(out) : "memory");
        else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(out) : "memory");
        else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(out) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(dst), "v"(out) : "memory");
    }
}
__global__ void prescale(const float* X, float* Xs, int64_t n, float c) { const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) Xs[i] = X[i] * c; }

struct Var { std::string name; std::function<void()> run; std::vector<float> ms; };
int main() {
    const int64_t N = 65536;
    float *X, *Xs, *K;
    hipMalloc(&X, N * QT * 4); hipMalloc(&Xs, N * QT * 4); hipMalloc(&K, N * N * 4);
    std::vector<float> hx(N * QT);
    srand(1);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 6.f - 3.f;
    hipMemcpy(X, hx.data(), N * QT * 4, hipMemcpyHostToDevice);
    prescale<<<(N * QT + 255) / 256, 256>>>(X, Xs, N * QT, 0.84932180028801904272f);
    hipDeviceSynchronize();
    const double gb = (double)N * N * 4 / 1e9;
    std::vector<Var> vs;
#define ADD(nm, TR, POL) vs.push_back({nm, [=] { gram_pol<TR, POL><<<dim3(N / 256, N / TR), 64>>>(Xs, K, N); }, {}})
    ADD("TR12 nt (production)", 12, 1);
    ADD("TR12 plain", 12, 0);
    ADD("TR12 sc0 sc1", 12, 2);
    ADD("TR12 sc1 nt", 12, 3);
    ADD("TR12 sc0 sc1 nt", 12, 4);
    ADD("TR12 sc1", 12, 5);
    ADD("TR12 sc0 nt", 12, 6);
    ADD("TR16 nt", 16, 1);
    ADD("TR16 sc0 sc1 nt", 16, 4);
    ADD("TR8 nt", 8, 1);
    ADD("TR8 sc0 sc1 nt", 8, 4);
    // spot check of one variant per policy: entry (17, 12345)
    for (auto& v : vs) {
        hipMemset(K, 0xff, 64 * N * 4);
        v.run(); hipDeviceSynchronize();
        float got; hipMemcpy(&got, K + 17 * N + 12345, 4, hipMemcpyDeviceToHost);
        double r2 = 0;
        for (int q = 0; q < QT; ++q) { const double d = (double)hx[17 * QT + q] - (double)hx[12345 * QT + q]; r2 += d * d; }
        printf("check %-24s %.6e vs %.6e\n", v.name.c_str(), got, exp(-0.5 * r2));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) for (auto& v : vs) v.run();
    hipDeviceSynchronize();
    const int rounds = 5, reps = 8;
    for (int rd = 0; rd < rounds; ++rd)
        for (auto& v : vs) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) v.run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            v.ms.push_back(ms / reps);
        }
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-24s min %7.3f med %7.3f ms -> %7.1f GB/s (min) %7.1f (med)\n", v.name.c_str(), v.ms[0], v.ms[v.ms.size() / 2], gb / v.ms[0] * 1e3, gb / v.ms[v.ms.size() / 2] * 1e3);
    }
    return 0;
}

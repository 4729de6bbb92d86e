// accuracy of the 1/sqrt used by the Cholesky kernels (hardware v_rsq_f64 + one third-order correction) against 1 / sqrt(d) in float64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* d, double* out, double* raw, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    out[i] = fma(y * e, fma(e, 0.375, 0.5), y);
    raw[i] = y;
}
int main() {
    const int n = 1 << 20;
    double* h = new double[n]; double* o = new double[n]; double* r = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = pow(10.0, -14.0 + 18.0 * u); }
    double *dd, *dout, *draw; hipMalloc(&dd, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&draw, n * 8);
    hipMemcpy(dd, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dd, dout, draw, n);
    hipMemcpy(o, dout, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r, draw, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_raw = 0;
    for (int i = 0; i < n; ++i) {
        const long double ref = 1.0L / sqrtl((long double)h[i]);
        const double e1 = fabs((double)(((long double)o[i] - ref) / ref)), e0 = fabs((double)(((long double)r[i] - ref) / ref));
        if (e1 > worst) worst = e1;
        if (e0 > worst_raw) worst_raw = e0;
    }
    printf("v_rsq_f64 alone: max rel err %.3e (2^%.1f); corrected: %.3e (%.2f ulp of 2^-53)\n", worst_raw, log2(worst_raw), worst, worst / 1.1102230246251565e-16);
    return 0;
}

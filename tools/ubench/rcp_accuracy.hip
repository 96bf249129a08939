// Accuracy of v_rcp_f64 (raw) and of rcp_approx (one Newton step) on gfx950: max relative error over random doubles.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/rcp_accuracy tools/ubench/rcp_accuracy.hip && tools/ubench/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *r0, double *r1, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    r1[i] = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> m(1.0, 2.0), e(-300, 300);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)e(g)) * ((i & 1) ? -1 : 1);
    double *dx, *d0, *d1;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    long double w0 = 0, w1 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        w0 = fmaxl(w0, fabsl(((long double)a[i] - t) / t));
        w1 = fmaxl(w1, fabsl(((long double)b[i] - t) / t));
    }
    std::printf("v_rcp_f64: max relative error %.3Le (2^%.1Lf);  + one Newton step: %.3Le (2^%.1Lf)\n", w0, log2l(w0), w1, log2l(w1));
    return 0;
}

// Cost of IEEE f64 division sequences in a single wave (development aid): hipcc -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void bench(double *io, unsigned long long *cyc, int reps) {
    double n[16], d[16], q[16];
    for (int i = 0; i < 16; ++i) { n[i] = io[threadIdx.x + 64 * i]; d[i] = io[threadIdx.x + 64 * (16 + i)]; q[i] = 0; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) q[i] += n[i] / d[i];
            if (MODE == 1) {  // the same Newton sequence without scaling and fix-up (valid in a safe exponent range)
                double x = __builtin_amdgcn_rcp(d[i]);
                x = __builtin_fma(__builtin_fma(-d[i], x, 1.0), x, x);
                x = __builtin_fma(__builtin_fma(-d[i], x, 1.0), x, x);
                const double q0 = n[i] * x;
                q[i] += __builtin_fma(__builtin_fma(-d[i], q0, n[i]), x, q0);
            }
            n[i] += 1e-9;  // keep the loop from being hoisted
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += q[i];
    io[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double *dio; unsigned long long *dc;
    (void)hipMalloc(&dio, 8 * 64 * 32); (void)hipMalloc(&dc, 8);
    std::vector<double> h(64 * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + 0.001 * (i % 977);
    const int reps = 200;
    for (int mode = 0; mode < 2; ++mode) {
        for (int k = 0; k < 2; ++k) {
            (void)hipMemcpy(dio, h.data(), 8 * h.size(), hipMemcpyHostToDevice);
            if (mode == 0) bench<0><<<1, 64>>>(dio, dc, reps); else bench<1><<<1, 64>>>(dio, dc, reps);
            unsigned long long c = 0;
            (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            double s; (void)hipMemcpy(&s, dio, 8, hipMemcpyDeviceToHost);
            if (k) printf("%s: %.1f cycles per division (16 independent, incl. one add each); checksum %.17g\n", mode ? "rcp + Newton, unscaled" : "compiler division", (double)c / (reps * 16.0), s);
        }
    }
    return 0;
}

// Single-wave issue / latency microbenchmarks for gfx950 (development aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O2 -o latency tools/ubench/latency.hip && ./latency
// One wave on one SIMD -- the regime of cert_solve_kernel (one wave per SIMD) -- executes R repetitions of an
// unrolled body of U instructions; cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int TEST>
__global__ void bench(double *out, unsigned long long *cyc, int reps) {
    double a = out[threadIdx.x], b = out[64 + threadIdx.x], c = out[128 + threadIdx.x], d = out[192 + threadIdx.x];
    double e = a + 1.0, f = b + 2.0, g = c + 3.0, h = d + 4.0;
    const double k1 = 1.0000001, k2 = 1e-9;
    int ia = threadIdx.x, ib = threadIdx.x * 3;
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 8 + 8) & 4095);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (TEST == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(k1), "v"(k2));) }
        if (TEST == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(k1), "v"(k2));) }
        if (TEST == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k1), "v"(k2));) }
        if (TEST == 3) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(k1));) }
        if (TEST == 4) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(k2));) }
        if (TEST == 5) { REP64(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(k2));) }
        if (TEST == 6) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia) : "v"(ib) : "vcc");) }
        if (TEST == 7) {  // cmp -> cndmask(2) -> cmp dependent chain (a running minimum; b changes so nothing folds)
            REP64(a = b < a ? b : a; asm volatile("" : "+v"(a), "+v"(b));) }
        if (TEST == 8) {  // two independent running minima interleaved
            REP64(a = b < a ? b : a; c = d < c ? d : c; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (TEST == 9) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
        if (TEST == 10) { REP64(asm volatile("v_rcp_f64 %0, %2\n v_rcp_f64 %1, %3" : "=v"(a), "=v"(b) : "v"(c), "v"(d));) }
        if (TEST == 11) { REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(ia));) }
        if (TEST == 12) { REP64(asm volatile("v_accvgpr_write_b32 a0, %1\n v_accvgpr_read_b32 %0, a0" : "=v"(ib) : "v"(ia) : "a0");) }
        if (TEST == 13) { REP64(asm volatile("s_or_b64 s[20:21], s[20:21], vcc\n s_and_b64 s[22:23], s[22:23], vcc" : : : "s20", "s21", "s22", "s23", "scc");) }
        if (TEST == 14) {  // VALU + independent SALU alternating
            REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_or_b64 s[20:21], s[20:21], vcc\n v_fma_f64 %1, %1, %2, %3\n s_and_b64 s[22:23], s[22:23], vcc" : "+v"(a), "+v"(b) : "v"(k1), "v"(k2) : "s20", "s21", "s22", "s23", "scc");) }
        if (TEST == 15) {  // (cmp & cmp) -> cndmask: mask logic on the scalar unit between compare and select
            REP64(a = ((b < a) & (c > d)) ? b : a; asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (TEST == 16) { REP64(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %2, %5\n v_add_f64 %3, %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k1), "v"(k2));) }
        if (TEST == 17) { REP64(asm volatile("v_writelane_b32 %0, s20, 3\n v_readlane_b32 s21, %0, 5" : "+v"(ia) : : "s20", "s21");) }
        if (TEST == 18) { REP64(asm volatile("v_mul_f64 %0, %0, %1\n v_add_f64 %0, %0, %2" : "+v"(a) : "v"(k1), "v"(k2));) }
        if (TEST == 19) {  // f32 dependent fma for comparison
            REP64(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(ia));) }
        if (TEST == 20) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(a) : "v"(k1), "v"(k2));) }
        if (TEST == 21) { REP64(asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2\n v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
        if (TEST == 22) {  // running minimum with index (cmp + 3 cndmask)
            REP64({ const bool lt = b < a; a = lt ? b : a; ia = lt ? ib : ia; } asm volatile("" : "+v"(a), "+v"(b), "+v"(ia), "+v"(ib));) }
        if (TEST == 23) { REP64(asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b) : "v"(ia));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d + e + f + g + h + ia + ib;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int T>
void run(const char *name, int per_rep, double *dout, unsigned long long *dcyc) {
    const int reps = 200;
    bench<T><<<1, 64>>>(dout, dcyc, reps);
    bench<T><<<1, 64>>>(dout, dcyc, reps);
    unsigned long long c = 0;
    hipError_t err = hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    if (err != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(err)); return; }
    printf("%-72s %7.2f cycles / instruction (%d per body)\n", name, (double)c / (reps * 64.0 * per_rep), per_rep);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    double *dout; unsigned long long *dcyc;
    hipMalloc(&dout, 8 * 256); hipMalloc(&dcyc, 8);
    std::vector<double> h(256, 1.25);
    hipMemcpy(dout, h.data(), 8 * 256, hipMemcpyHostToDevice);
    run<0>("v_fma_f64 dependent chain", 1, dout, dcyc);
    run<1>("v_fma_f64 2 independent chains", 2, dout, dcyc);
    run<2>("v_fma_f64 4 independent chains", 4, dout, dcyc);
    run<3>("v_mul_f64 dependent", 1, dout, dcyc);
    run<4>("v_add_f64 dependent", 1, dout, dcyc);
    run<5>("v_max_f64 dependent", 1, dout, dcyc);
    run<6>("v_cndmask_b32 dependent", 1, dout, dcyc);
    run<7>("running min: cmp + 2 cndmask, dependent", 3, dout, dcyc);
    run<8>("two running minima interleaved", 6, dout, dcyc);
    run<9>("v_rcp_f64 dependent", 1, dout, dcyc);
    run<10>("v_rcp_f64 independent", 2, dout, dcyc);
    run<13>("s_or_b64 / s_and_b64", 2, dout, dcyc);
    run<14>("v_fma_f64 (2 chains) alternating with SALU", 4, dout, dcyc);
    run<15>("2 cmp -> s_and -> 2 cndmask", 5, dout, dcyc);
    run<16>("2 mul + 2 add independent", 4, dout, dcyc);
    run<18>("mul -> add dependent pair", 2, dout, dcyc);
    run<19>("v_fma_f32 dependent", 1, dout, dcyc);
    run<20>("v_fma_f64 dependent + s_nop 0", 2, dout, dcyc);
    run<21>("v_div_scale_f64 -> v_div_fmas_f64", 2, dout, dcyc);
    run<22>("running min with index: cmp + 3 cndmask", 4, dout, dcyc);
    run<11>("ds_read_b32 dependent + wait", 1, dout, dcyc);
    run<12>("v_accvgpr write/read", 2, dout, dcyc);
    run<17>("v_writelane / v_readlane", 2, dout, dcyc);
    run<23>("2 ds_read_b64 + wait", 2, dout, dcyc);
    return 0;
}

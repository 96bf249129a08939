// tpr_kernels.hip -- libtoppra_hip.so: HIP kernels + the C-ABI of include/toppra_hip.h (gfx950).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
// -ffp-contract=off is a correctness flag, not a tuning knob: see tpr_device.hpp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/toppra_hip.h"
#include "tpr_device.hpp"
#include "tpr_lane.hip.inc"
#include "tpr_group.hip.inc"
#include "tpr_wave.hip.inc"
#include "tpr_pair.hip.inc"
#include "tpr_spline.hip.inc"
#include "tpr_param.hip.inc"
#include "tpr_robust_args.hpp"
#include "tpr_dense_args.hpp"

#define TPR_TU_CAT3_(a, b) a##b
#define TPR_TU_CAT3(a, b) TPR_TU_CAT3_(a, b)
// kernel family 3, one translation unit per dof (tpr_cert_tu.hip): 1..TPR_CERT_MAX_DOF (build.py: 15)
#ifndef TPR_CERT_MAX_DOF
#define TPR_CERT_MAX_DOF 15
#endif
#ifdef TPR_SINGLE_TU  // development builds with instrumentation: everything in this translation unit, ONE dof for family 3 (7; -DTPR_SINGLE_TU_D=<dof>)
#ifndef TPR_SINGLE_TU_D
#define TPR_SINGLE_TU_D 7
#endif
#define TPR_TU_D TPR_SINGLE_TU_D
#include "tpr_cert_tu.hip"
#undef TPR_TU_D
#define TPR_TU_HALF 2
#include "tpr_robust_tu.hip"
#undef TPR_TU_HALF
#include "tpr_dense_tu.hip"
#else
extern "C" {
__attribute__((visibility("hidden"))) int tpr_tu_dense_launch(const tpr::DenseArgs *, int, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_robust_launch_lo(const tpr::RobustArgs *, size_t, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_robust_launch_hi(const tpr::RobustArgs *, size_t, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_robust_lane_launch(const tpr::RobustArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_1(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_1(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_1(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_2(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_2(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_2(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_3(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_3(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_3(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_4(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_4(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_4(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_5(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_5(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_5(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_6(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_6(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_6(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_7(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_7(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_7(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_8(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_8(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_8(const tpr::GroupArgs *, hipStream_t);
#if TPR_CERT_MAX_DOF >= 9
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_9(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_9(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_9(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 10
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_10(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_10(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_10(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 11
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_11(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_11(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_11(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 12
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_12(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_12(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_12(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 13
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_13(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_13(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_13(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 14
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_14(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_14(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_14(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 15
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_15(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_15(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_15(const tpr::GroupArgs *, hipStream_t);
#endif
#if TPR_CERT_MAX_DOF >= 16
__attribute__((visibility("hidden"))) int tpr_tu_cert_launch_16(const tpr::GroupArgs *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_feasible_launch_16(const tpr::GroupArgs *, double *, hipStream_t);
__attribute__((visibility("hidden"))) int tpr_tu_cert_sd_launch_16(const tpr::GroupArgs *, hipStream_t);
#endif
}
#endif

namespace {

thread_local std::string g_err;
// Default device of host-pointer calls: the calling THREAD's last successful tpr_init, else the process's (a thread
// that never called tpr_init inherits what another one selected).  Both, and the table of verified devices, may be
// touched by several threads working on different GPUs.
std::atomic<int> g_process_device{-1};
thread_local int t_device = -1;
std::atomic<bool> g_checked[64];  // devices already verified to be gfx950
inline int default_device() { return t_device >= 0 ? t_device : g_process_device.load(std::memory_order_relaxed); }
#define g_device default_device()

// HIP's current device is per thread and other libraries (torch) move it.  Every entry point runs on
// the device its data lives on -- the device of the pointers with TPR_DEVICE_PTRS, the tpr_init()
// device otherwise -- and puts the caller's current device back on return.
struct DeviceScope {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int want) : dev(want) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceScope() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

// Device a call should run on: the one `device_ptr` lives on (TPR_DEVICE_PTRS), else g_device.
int call_device(bool device_ptrs, const void *device_ptr) {
    if (device_ptrs && device_ptr) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, device_ptr) == hipSuccess && attr.device >= 0) return attr.device;
        (void)hipGetLastError();
    }
    return g_device;
}

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(TPR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

// Host<->device staging for callers that hand over host buffers.  With TPR_DEVICE_PTRS every
// pointer passes through untouched and nothing here allocates.
struct Staging {
    bool device_ptrs;
    hipStream_t stream;
    std::vector<void *> owned;
    struct Out { void *host; void *dev; size_t bytes; };
    std::vector<Out> outs;
    hipError_t err = hipSuccess;

    Staging(bool dev, hipStream_t s) : device_ptrs(dev), stream(s) {}
    ~Staging() {
        for (void *p : owned) (void)hipFreeAsync(p, stream);  // stream-ordered pool: no device sync, memory is reused
    }
    template <class T>
    const T *in(const T *p, size_t count) {
        if (!p || device_ptrs || count == 0) return p;
        void *d = nullptr;
        if (err == hipSuccess) err = hipMallocAsync(&d, count * sizeof(T), stream);
        if (err != hipSuccess) return nullptr;
        owned.push_back(d);
        err = hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, stream);
        return static_cast<const T *>(d);
    }
    template <class T>
    T *out(T *p, size_t count, bool copy_in = false) {
        if (!p || device_ptrs || count == 0) return p;
        void *d = nullptr;
        if (err == hipSuccess) err = hipMallocAsync(&d, count * sizeof(T), stream);
        if (err != hipSuccess) return nullptr;
        owned.push_back(d);
        if (copy_in) err = hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, stream);
        outs.push_back({p, d, count * sizeof(T)});
        return static_cast<T *>(d);
    }
    hipError_t finish() {
        if (err != hipSuccess) return err;
        // a kernel that could not be launched leaves its error in the runtime's per-thread slot only: ask for it on
        // both paths (round 3: a host-buffer call whose launch failed returned uninitialised outputs without a word)
        const hipError_t launch = hipGetLastError();
        if (launch != hipSuccess || device_ptrs) return launch;
        for (auto &o : outs) {
            err = hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, stream);
            if (err != hipSuccess) return err;
        }
        return hipStreamSynchronize(stream);
    }
};

int check_problem(const tpr_problem *p) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p) return fail(TPR_E_BADARG, "null problem");
    if (p->B < 0 || p->N < 1 || p->nseg < 1) return fail(TPR_E_BADARG, "need B >= 0, N >= 1, nseg >= 1");
    if (p->d < 1 || p->d > TPR_MAX_DOF) return fail(TPR_E_UNSUPPORTED, "dof must be in [1, TPR_MAX_DOF]");
    if (!p->coef || !p->breaks || !p->grid) return fail(TPR_E_BADARG, "coef/breaks/grid are required");
    if ((p->flags & TPR_HAS_VELOCITY) && !p->vlim) return fail(TPR_E_BADARG, "TPR_HAS_VELOCITY without vlim");
    if ((p->flags & TPR_HAS_ACCELERATION) && !p->alim) return fail(TPR_E_BADARG, "TPR_HAS_ACCELERATION without alim");
    return TPR_E_OK;
}

int rows_per_lp(const tpr_problem *p) {
    return 2 + ((p->flags & TPR_HAS_ACCELERATION) ? ((p->flags & TPR_ACC_INTERPOLATION) ? 4 : 2) * p->d : 0);
}

// Stage the inputs of a problem; returns the kernel argument block.
tpr::BatchArgs stage_problem(const tpr_problem *p, Staging &S) {
    tpr::BatchArgs A{};
    const size_t B = (size_t)p->B, d = (size_t)p->d, nseg = (size_t)p->nseg, N = (size_t)p->N;
    A.B = p->B; A.d = p->d; A.nseg = p->nseg; A.N = p->N; A.flags = p->flags;
    A.coef = S.in(p->coef, B * 4 * nseg * d);
    A.breaks = S.in(p->breaks, ((p->flags & TPR_BREAKS_PER_TRAJ) ? B : 1) * (nseg + 1));
    A.grid = S.in(p->grid, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1));
    A.vlim = S.in(p->vlim, B * d * 2);
    A.alim = S.in(p->alim, B * d * 2);
    A.sd_start = S.in(p->sd_start, B);
    A.sd_end = S.in(p->sd_end, B);
    return A;
}

constexpr size_t kMaxDynamicLds = 64 * 1024;
#ifndef TPR_PAIR_AUTO_MIN_BATCH
#define TPR_PAIR_AUTO_MIN_BATCH 2560   // auto: two trajectories per wave between these batch sizes (tools/gpu_crossover.py)
#endif
#ifndef TPR_PAIR_AUTO_MAX_BATCH
#define TPR_PAIR_AUTO_MAX_BATCH 9215
#endif
#ifndef TPR_WAVE_AUTO_MAX_BATCH
#define TPR_WAVE_AUTO_MAX_BATCH 5120  // auto: one wave per trajectory up to this many trajectories (4096: 0.86 vs 1.11 ms for
                                      // family 2; 8192: 1.51 vs 1.31 -- tools/gpu_wave_check.py)
#endif

template <int D, int L>
size_t group_lds_bytes(int nseg, int threads, bool table_in_lds) {
    return (size_t)(threads / L) * tpr::GroupCfg<D, L>::lds_doubles(nseg, table_in_lds) * sizeof(double);
}

// Launch geometry of the rows-across-lanes kernel.  The spline table is staged in LDS when a
// 64-thread block's worth fits in 64 KB (always for ordinary waypoint counts); otherwise it stays
// in global memory.  Block size: the largest of 256 / 128 / 64 threads that fits, shrunk further
// while the batch would leave CUs idle (256 CUs; small batches such as BASELINE config 2's 4096
// trajectories only make 128 blocks of 256 threads).
template <int D, int L>
int launch_group(const tpr::BatchArgs &A, hipStream_t stream) {
    const bool table_in_lds = group_lds_bytes<D, L>(A.nseg, 64, true) <= kMaxDynamicLds;
    int threads = 64;
    for (int t = 256; t > 64; t /= 2)
        if (group_lds_bytes<D, L>(A.nseg, t, table_in_lds) <= kMaxDynamicLds) { threads = t; break; }
    while (threads > 64 && (long long)A.B * L / threads < 4 * 256) threads /= 2;
    tpr::GroupArgs G{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim,
                     A.sd_start, A.sd_end, A.sd2, A.sd, A.u, A.K, A.status, A.sd_end_hi, A.backward_only};
    const int groups = threads / L;
    size_t lds = group_lds_bytes<D, L>(A.nseg, threads, table_in_lds);
#ifdef TPR_LDS_PAD_EXPERIMENT  // occupancy experiments (development builds only)
    if (const char *pad = std::getenv("TPR_LDS_PAD")) lds += (size_t)std::atoi(pad);
#endif
    const dim3 grid((A.B + groups - 1) / groups), block(threads);
    if (table_in_lds) hipLaunchKernelGGL((tpr::group_solve_kernel<D, L, true>), grid, block, lds, stream, G);
    else hipLaunchKernelGGL((tpr::group_solve_kernel<D, L, false>), grid, block, lds, stream, G);
    return TPR_E_OK;
}

// The rows-across-lanes kernels cover every constraint set of the path -- acceleration with
// Interpolation (the reference's default) or Collocation or absent, velocity optional -- for every
// supported dof: 8 lanes per trajectory up to d = 8, 16 lanes above.  (Missing acceleration blocks are
// disabled rows in the Interpolation slot layout, see GroupTraj::nblk.)
bool group_supported(const tpr::BatchArgs &A) { return A.d >= 1 && A.d <= TPR_MAX_DOF_FAST; }

// ... the robust kernel and family 3 are written for the full Interpolation row set
bool interp_rows(const tpr::BatchArgs &A) {
    const int need = TPR_HAS_ACCELERATION | TPR_ACC_INTERPOLATION;
    return (A.flags & need) == need;
}

template <int D, int L>
int launch_sd_forward(const tpr::SdArgs &A, hipStream_t stream) {
    const bool table_in_lds = group_lds_bytes<D, L>(A.nseg, 64, true) <= kMaxDynamicLds;
    int threads = 64;
    for (int t = 256; t > 64; t /= 2)
        if (group_lds_bytes<D, L>(A.nseg, t, table_in_lds) <= kMaxDynamicLds) { threads = t; break; }
    while (threads > 64 && (long long)A.B * L / threads < 4 * 256) threads /= 2;
    const int groups = threads / L;
    const size_t lds = group_lds_bytes<D, L>(A.nseg, threads, table_in_lds);
    const dim3 grid((A.B + groups - 1) / groups), block(threads);
    if (table_in_lds) hipLaunchKernelGGL((tpr::group_sd_forward_kernel<D, L, true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((tpr::group_sd_forward_kernel<D, L, false>), grid, block, lds, stream, A);
    return TPR_E_OK;
}

int dispatch_sd_forward(int d, const tpr::SdArgs &A, hipStream_t stream) {
    switch (d) {
        case 1: return launch_sd_forward<1, 8>(A, stream);
        case 2: return launch_sd_forward<2, 8>(A, stream);
        case 3: return launch_sd_forward<3, 8>(A, stream);
        case 4: return launch_sd_forward<4, 8>(A, stream);
        case 5: return launch_sd_forward<5, 8>(A, stream);
        case 6: return launch_sd_forward<6, 8>(A, stream);
        case 7: return launch_sd_forward<7, 8>(A, stream);
        case 8: return launch_sd_forward<8, 8>(A, stream);
        case 9: return launch_sd_forward<9, 16>(A, stream);
        case 10: return launch_sd_forward<10, 16>(A, stream);
        case 11: return launch_sd_forward<11, 16>(A, stream);
        case 12: return launch_sd_forward<12, 16>(A, stream);
        case 13: return launch_sd_forward<13, 16>(A, stream);
        case 14: return launch_sd_forward<14, 16>(A, stream);
        case 15: return launch_sd_forward<15, 16>(A, stream);
        case 16: return launch_sd_forward<16, 16>(A, stream);
    }
    return fail(TPR_E_UNSUPPORTED, "dof out of range");
}

// The certified lane kernel (family 3) serves the same constraint set up to 8 dof when sd2, u and
// status are requested; the strict mode stays with family 2.
bool cert_supported(const tpr::BatchArgs &A) {
    return group_supported(A) && (A.flags & TPR_HAS_ACCELERATION) && A.d <= TPR_CERT_MAX_DOF &&
           !(A.flags & TPR_STRICT_SEIDEL) && A.N >= 1 && !A.active &&
           (A.backward_only || (A.sd2 && A.u && A.status));
}

// Return codes of the per-dof translation units of family 3: 0, or a negative "this instantiation does not exist".
int cert_tu_rc(int rc) {
    return rc < 0 ? fail(TPR_E_UNSUPPORTED, "kernel family 3: this combination of dof / discretisation / certificate mode is not instantiated") : rc;
}

// Kernel family 3 lives in its own translation units, one per dof (tpr_cert_tu.hip; build.py compiles them in parallel).
int launch_cert(const tpr::BatchArgs &A, hipStream_t stream) {
    tpr::GroupArgs G{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim,
                     A.sd_start, A.sd_end, A.sd2, A.sd, A.u, A.K, A.status, A.sd_end_hi, A.backward_only};
    switch (A.d) {
#ifndef TPR_CERT_DEV  // development builds instantiate 7 dof only
        case 1: return cert_tu_rc(tpr_tu_cert_launch_1(&G, stream));
        case 2: return cert_tu_rc(tpr_tu_cert_launch_2(&G, stream));
        case 3: return cert_tu_rc(tpr_tu_cert_launch_3(&G, stream));
        case 4: return cert_tu_rc(tpr_tu_cert_launch_4(&G, stream));
        case 5: return cert_tu_rc(tpr_tu_cert_launch_5(&G, stream));
        case 6: return cert_tu_rc(tpr_tu_cert_launch_6(&G, stream));
        case 8: return cert_tu_rc(tpr_tu_cert_launch_8(&G, stream));
#if TPR_CERT_MAX_DOF >= 9
        case 9: return cert_tu_rc(tpr_tu_cert_launch_9(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 10
        case 10: return cert_tu_rc(tpr_tu_cert_launch_10(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 11
        case 11: return cert_tu_rc(tpr_tu_cert_launch_11(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 12
        case 12: return cert_tu_rc(tpr_tu_cert_launch_12(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 13
        case 13: return cert_tu_rc(tpr_tu_cert_launch_13(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 14
        case 14: return cert_tu_rc(tpr_tu_cert_launch_14(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 15
        case 15: return cert_tu_rc(tpr_tu_cert_launch_15(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 16
        case 16: return cert_tu_rc(tpr_tu_cert_launch_16(&G, stream));
#endif
        case 7: return cert_tu_rc(tpr_tu_cert_launch_7(&G, stream));
#else
        case TPR_SINGLE_TU_D: return TPR_TU_CAT3(tpr_tu_cert_launch_, TPR_SINGLE_TU_D)(&G, stream);
#endif
    }
    return fail(TPR_E_UNSUPPORTED, "variant 3: dof not instantiated");
}

// compute_feasible_sets on the certified lane design: the constraint sets and dofs of family 3, fresh warm-start state
bool cert_feasible_supported(const tpr::BatchArgs &A) {
    return group_supported(A) && (A.flags & TPR_HAS_ACCELERATION) && A.d <= TPR_CERT_MAX_DOF &&
           !(A.flags & TPR_STRICT_SEIDEL) && !A.active;
}

int launch_cert_feasible(const tpr::BatchArgs &A, double *X, hipStream_t stream) {
    tpr::GroupArgs G{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim,
                     A.sd_start, A.sd_end, A.sd2, A.sd, A.u, A.K, A.status};
    switch (A.d) {
#ifndef TPR_CERT_DEV
        case 1: return cert_tu_rc(tpr_tu_cert_feasible_launch_1(&G, X, stream));
        case 2: return cert_tu_rc(tpr_tu_cert_feasible_launch_2(&G, X, stream));
        case 3: return cert_tu_rc(tpr_tu_cert_feasible_launch_3(&G, X, stream));
        case 4: return cert_tu_rc(tpr_tu_cert_feasible_launch_4(&G, X, stream));
        case 5: return cert_tu_rc(tpr_tu_cert_feasible_launch_5(&G, X, stream));
        case 6: return cert_tu_rc(tpr_tu_cert_feasible_launch_6(&G, X, stream));
        case 8: return cert_tu_rc(tpr_tu_cert_feasible_launch_8(&G, X, stream));
#if TPR_CERT_MAX_DOF >= 9
        case 9: return cert_tu_rc(tpr_tu_cert_feasible_launch_9(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 10
        case 10: return cert_tu_rc(tpr_tu_cert_feasible_launch_10(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 11
        case 11: return cert_tu_rc(tpr_tu_cert_feasible_launch_11(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 12
        case 12: return cert_tu_rc(tpr_tu_cert_feasible_launch_12(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 13
        case 13: return cert_tu_rc(tpr_tu_cert_feasible_launch_13(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 14
        case 14: return cert_tu_rc(tpr_tu_cert_feasible_launch_14(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 15
        case 15: return cert_tu_rc(tpr_tu_cert_feasible_launch_15(&G, X, stream));
#endif
#if TPR_CERT_MAX_DOF >= 16
        case 16: return cert_tu_rc(tpr_tu_cert_feasible_launch_16(&G, X, stream));
#endif
        case 7: return cert_tu_rc(tpr_tu_cert_feasible_launch_7(&G, X, stream));
#else
        case TPR_SINGLE_TU_D: return TPR_TU_CAT3(tpr_tu_cert_feasible_launch_, TPR_SINGLE_TU_D)(&G, X, stream);
#endif
    }
    return fail(TPR_E_UNSUPPORTED, "variant 3: dof not instantiated");
}

// TOPPRAsd on family 3: backward scan and both forward profiles in one launch
int launch_cert_sd(const tpr::BatchArgs &A, double *xf, double *uf, double *xl, double *ul, double *dur, hipStream_t stream) {
    tpr::GroupArgs G{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim,
                     A.sd_start, A.sd_end, A.sd2, nullptr, A.u, A.K, A.status, nullptr, 0, xf, uf, xl, ul};
    G.sd_dur = dur;
    switch (A.d) {
#ifndef TPR_CERT_DEV
        case 1: return cert_tu_rc(tpr_tu_cert_sd_launch_1(&G, stream));
        case 2: return cert_tu_rc(tpr_tu_cert_sd_launch_2(&G, stream));
        case 3: return cert_tu_rc(tpr_tu_cert_sd_launch_3(&G, stream));
        case 4: return cert_tu_rc(tpr_tu_cert_sd_launch_4(&G, stream));
        case 5: return cert_tu_rc(tpr_tu_cert_sd_launch_5(&G, stream));
        case 6: return cert_tu_rc(tpr_tu_cert_sd_launch_6(&G, stream));
        case 8: return cert_tu_rc(tpr_tu_cert_sd_launch_8(&G, stream));
#if TPR_CERT_MAX_DOF >= 9
        case 9: return cert_tu_rc(tpr_tu_cert_sd_launch_9(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 10
        case 10: return cert_tu_rc(tpr_tu_cert_sd_launch_10(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 11
        case 11: return cert_tu_rc(tpr_tu_cert_sd_launch_11(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 12
        case 12: return cert_tu_rc(tpr_tu_cert_sd_launch_12(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 13
        case 13: return cert_tu_rc(tpr_tu_cert_sd_launch_13(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 14
        case 14: return cert_tu_rc(tpr_tu_cert_sd_launch_14(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 15
        case 15: return cert_tu_rc(tpr_tu_cert_sd_launch_15(&G, stream));
#endif
#if TPR_CERT_MAX_DOF >= 16
        case 16: return cert_tu_rc(tpr_tu_cert_sd_launch_16(&G, stream));
#endif
        case 7: return cert_tu_rc(tpr_tu_cert_sd_launch_7(&G, stream));
#else
        case TPR_SINGLE_TU_D: return TPR_TU_CAT3(tpr_tu_cert_sd_launch_, TPR_SINGLE_TU_D)(&G, stream);
#endif
    }
    return fail(TPR_E_UNSUPPORTED, "variant 3: dof not instantiated");
}

template <int D, int L>
int launch_group_feasible(const tpr::BatchArgs &A, double *X, hipStream_t stream) {
    const bool table_in_lds = group_lds_bytes<D, L>(A.nseg, 64, true) <= kMaxDynamicLds;
    int threads = 64;
    for (int t = 256; t > 64; t /= 2)
        if (group_lds_bytes<D, L>(A.nseg, t, table_in_lds) <= kMaxDynamicLds) { threads = t; break; }
    while (threads > 64 && (long long)A.B * L / threads < 4 * 256) threads /= 2;
    tpr::GroupArgs G{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim,
                     A.sd_start, A.sd_end, A.sd2, A.sd, A.u, A.K, A.status};
    const int groups = threads / L;
    const size_t lds = group_lds_bytes<D, L>(A.nseg, threads, table_in_lds);
    const dim3 grid((A.B + groups - 1) / groups), block(threads);
    if (table_in_lds) hipLaunchKernelGGL((tpr::group_feasible_kernel<D, L, true>), grid, block, lds, stream, G, X);
    else hipLaunchKernelGGL((tpr::group_feasible_kernel<D, L, false>), grid, block, lds, stream, G, X);
    return TPR_E_OK;
}

int dispatch_group_feasible(const tpr::BatchArgs &A, double *X, hipStream_t stream) {
    switch (A.d) {
        case 1: return launch_group_feasible<1, 8>(A, X, stream);
        case 2: return launch_group_feasible<2, 8>(A, X, stream);
        case 3: return launch_group_feasible<3, 8>(A, X, stream);
        case 4: return launch_group_feasible<4, 8>(A, X, stream);
        case 5: return launch_group_feasible<5, 8>(A, X, stream);
        case 6: return launch_group_feasible<6, 8>(A, X, stream);
        case 7: return launch_group_feasible<7, 8>(A, X, stream);
        case 8: return launch_group_feasible<8, 8>(A, X, stream);
        case 9: return launch_group_feasible<9, 16>(A, X, stream);
        case 10: return launch_group_feasible<10, 16>(A, X, stream);
        case 11: return launch_group_feasible<11, 16>(A, X, stream);
        case 12: return launch_group_feasible<12, 16>(A, X, stream);
        case 13: return launch_group_feasible<13, 16>(A, X, stream);
        case 14: return launch_group_feasible<14, 16>(A, X, stream);
        case 15: return launch_group_feasible<15, 16>(A, X, stream);
        case 16: return launch_group_feasible<16, 16>(A, X, stream);
    }
    return fail(TPR_E_UNSUPPORTED, "dof out of range");
}

// The robust (conic) kernels live in their own translation units (tpr_robust_tu.hip): 0 = launched, 1 = the shape needs
// the generic lane kernel (very long spline tables), -1 = dof not served.
int dispatch_group_robust(const tpr::RobustArgs &P, hipStream_t stream) {
    int rc = P.A.d <= 8 ? tpr_tu_robust_launch_lo(&P, kMaxDynamicLds, stream) : tpr_tu_robust_launch_hi(&P, kMaxDynamicLds, stream);
    if (rc == 1) rc = tpr_tu_robust_lane_launch(&P, stream);
    return rc < 0 ? fail(TPR_E_UNSUPPORTED, "dof out of range") : TPR_E_OK;
}

// Family 4 (one trajectory per wave): every constraint set, every dof; grid, x box and K of the trajectory must
// fit the dynamic LDS of a block (5 (N+1) doubles: N <= 1480; the spline table joins them when there is room).
size_t wave_lds_bytes(const tpr::BatchArgs &A, bool table_in_lds) {
    return tpr::wave_lds_doubles(A.N, A.nseg, A.d, table_in_lds) * sizeof(double);
}
bool wave_supported(const tpr::BatchArgs &A) {
    return A.d >= 1 && A.d <= TPR_MAX_DOF && A.N >= 1 && A.nseg <= 65535 && wave_lds_bytes(A, false) <= kMaxDynamicLds;
}

#ifndef TPR_WAVE_SPLIT_MAX_BATCH
#define TPR_WAVE_SPLIT_MAX_BATCH 768  // two waves per trajectory (one per LP of a backward stage) up to this many trajectories
#endif
int launch_wave(const tpr::BatchArgs &A, hipStream_t stream) {
    const bool table = wave_lds_bytes(A, true) <= kMaxDynamicLds;
    const size_t lds = wave_lds_bytes(A, table);
    const int slots = (4 * A.d + 6 + 63) / 64;  // virtual rows per LP / 64 lanes
    // a handful of trajectories (BASELINE config 1): the two LPs of a backward stage on two waves (wave_solve_kernel<.., SPLIT>)
    const bool split = slots == 1 && !A.feasible_X && A.B <= TPR_WAVE_SPLIT_MAX_BATCH;
    const dim3 grid(A.B), block(split ? 128 : 64);
#define TPR_LAUNCH_WAVE(SS)                                                                                   \
    do {                                                                                                      \
        if (table) hipLaunchKernelGGL((tpr::wave_solve_kernel<SS, true>), grid, block, lds, stream, A);       \
        else hipLaunchKernelGGL((tpr::wave_solve_kernel<SS, false>), grid, block, lds, stream, A);            \
    } while (0)
    if (split) {
        if (table) hipLaunchKernelGGL((tpr::wave_solve_kernel<1, true, true>), grid, block, lds, stream, A);
        else hipLaunchKernelGGL((tpr::wave_solve_kernel<1, false, true>), grid, block, lds, stream, A);
        return TPR_E_OK;
    }
    switch (slots) {
        case 1: TPR_LAUNCH_WAVE(1); break;
        case 2: TPR_LAUNCH_WAVE(2); break;
        case 3: TPR_LAUNCH_WAVE(3); break;
        default: return fail(TPR_E_UNSUPPORTED, "variant 4: dof out of range");
    }
#undef TPR_LAUNCH_WAVE
    return TPR_E_OK;
}

// Family 5 (two trajectories per wave, 32 lanes each): the fused solve for 1..7 dof; both trajectories' LDS tables in one
// block.  Not the wrapper's warm-start state, not the stand-alone backward scan, not feasible sets (family 4 has those).
size_t pair_lds_bytes(const tpr::BatchArgs &A, bool table_in_lds) {
    return 2 * ((tpr::wave_lds_doubles(A.N, A.nseg, A.d, table_in_lds) + 1) & ~(size_t)1) * sizeof(double);
}
bool pair_supported(const tpr::BatchArgs &A) {
    return A.d >= 1 && A.d <= 7 && A.N >= 1 && A.nseg <= 65535 && !A.active && !A.feasible_X && !A.backward_only && !A.sd_end_hi &&
           pair_lds_bytes(A, false) <= kMaxDynamicLds;
}
int launch_pair(const tpr::BatchArgs &A, hipStream_t stream) {
    const bool table = pair_lds_bytes(A, true) <= kMaxDynamicLds;
    const size_t lds = pair_lds_bytes(A, table);
    const dim3 grid((A.B + 1) / 2), block(64);
    if (table) hipLaunchKernelGGL((tpr::pair_solve_kernel<true>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((tpr::pair_solve_kernel<false>), grid, block, lds, stream, A);
    return TPR_E_OK;
}

// Batch size from which family 3 is the automatic choice (solve, TOPPRAsd).
// (measured against family 2, 9..14 dof: tools/gpu_crossover_slim.py, profiles/r06_crossover_9_14_dof.log)
int cert_auto_from(int d) { return d <= 8 ? 9216 : (d <= 10 ? 14336 : (d == 11 ? 15360 : (d == 12 ? 17408 : (d == 13 ? 22528 : (d == 14 ? 27648 : 36864))))); }

// Kernel family of a solve (tpr_problem.variant 0 = auto).
int pick_variant(int requested, const tpr::BatchArgs &A) {
    if (requested != 0) return requested;
    // the wrapper object's warm-start state in / out: family 4 maintains it, and so does the generic lane kernel (family 1)
    // for what family 4 cannot take (N > 1480); families 2 and 3 neither read nor update it
    if (A.active) return wave_supported(A) ? 4 : 1;
    // Batches that cannot fill the chip are bound by the latency of a trajectory's 3N sequential stage LPs: one
    // wave per trajectory (family 4).  Family 3 finishes up to 65536 trajectories (one wave per SIMD) in one
    // fixed-latency round, which beats family 2's throughput from about a quarter of that batch upward
    // (tools/gpu_crossover.py); family 2 serves the strict mode and what is left.
    // ... two trajectories per wave (family 5) from the batch size at which one wave per trajectory stops being free: the
    // chip holds 1024 waves at one per SIMD, and family 4's waves leave half their lanes idle at <= 7 dof
    // (... while eight blocks still fit a CU's LDS -- two waves per SIMD, all a 4096-trajectory batch can use: N <= ~240 at 7 dof)
    if (pair_supported(A) && A.B >= TPR_PAIR_AUTO_MIN_BATCH && A.B <= TPR_PAIR_AUTO_MAX_BATCH && pair_lds_bytes(A, true) <= 160 * 1024 / 8) return 5;
    if (wave_supported(A) && A.B <= TPR_WAVE_AUTO_MAX_BATCH) return 4;
    // (round 4, after families 2 and 4 learnt to follow the lower-bound trace too: at 7 dof family 2 leads between ~5600 and
    // ~9200 trajectories, 1.7 - 1.9 ms against family 3's 2.1 - 2.2 at any size up to 65536; the slim blocks of 9..13 dof
    // take 3.2 - 4.4 ms for a partial round and pay from ~18000 / ~22000 trajectories: profiles/r04_family_crossover.log)
    if (cert_supported(A) && A.B >= cert_auto_from(A.d)) return 3;
    return group_supported(A) ? 2 : (wave_supported(A) ? 4 : 1);
}

int launch_solve(const tpr_problem *p, const tpr::BatchArgs &A, hipStream_t stream) {
    if (A.B == 0) return TPR_E_OK;
    const int variant = pick_variant(p->variant, A);
    if (A.active && (variant == 2 || variant == 3))
        return fail(TPR_E_UNSUPPORTED, "tpr_problem.active (warm-start state in / out) is maintained by kernel families 4 and 1 only: leave variant at 0");
    switch (variant) {
        case 5: {
            if (!pair_supported(A))
                return fail(TPR_E_UNSUPPORTED, "variant 5 (two trajectories per wave) serves the fused solve for 1..7 dof with N <= ~800 and no warm-start state");
            return launch_pair(A, stream);
        }
        case 4: {
            if (!wave_supported(A)) return fail(TPR_E_UNSUPPORTED, "variant 4: N too large for the per-trajectory LDS tables (N <= 1480)");
            return launch_wave(A, stream);
        }
        case 3: {
            if (!cert_supported(A))
                return fail(TPR_E_UNSUPPORTED, "variant 3 needs an acceleration constraint, d <= 15, sd2/u/status outputs, no strict mode");
            return launch_cert(A, stream);
        }
        case 2: {
            if (!group_supported(A)) return fail(TPR_E_UNSUPPORTED, "variant 2: dof out of range");
            // up to 8 dof a trajectory fits 8 lanes; batches that leave most SIMDs idle at that width
            // (<= 8192 trajectories = 1024 waves) run 16 lanes per trajectory: 1.42 -> 1.15 ms at 4096 x 7 x 200
            const bool wide = A.B <= 8192;
#ifdef TPR_WIDE_EXPERIMENT  // lanes per trajectory for small batches (development builds): TPR_LANES=32|64
            if (const char *e = std::getenv("TPR_LANES")) {
                const int L = std::atoi(e);
                if (A.d == 7 && L == 32) return launch_group<7, 32>(A, stream);
                if (A.d == 7 && L == 64) return launch_group<7, 64>(A, stream);
            }
#endif
            switch (A.d) {
#define TPR_GROUP_CASE(DD) case DD: return wide ? launch_group<DD, 16>(A, stream) : launch_group<DD, 8>(A, stream)
                TPR_GROUP_CASE(1);
                TPR_GROUP_CASE(2);
                TPR_GROUP_CASE(3);
                TPR_GROUP_CASE(4);
                TPR_GROUP_CASE(5);
                TPR_GROUP_CASE(6);
                TPR_GROUP_CASE(7);
                TPR_GROUP_CASE(8);
#undef TPR_GROUP_CASE
                case 9: return launch_group<9, 16>(A, stream);
                case 10: return launch_group<10, 16>(A, stream);
                case 11: return launch_group<11, 16>(A, stream);
                case 12: return launch_group<12, 16>(A, stream);
                case 13: return launch_group<13, 16>(A, stream);
                case 14: return launch_group<14, 16>(A, stream);
                case 15: return launch_group<15, 16>(A, stream);
                case 16: return launch_group<16, 16>(A, stream);
            }
            return TPR_E_OK;
        }
        case 1: {
            const int block = 64;
            if (A.backward_only)  // compute_controllable_sets on its own
                hipLaunchKernelGGL(tpr::lane_controllable_kernel, dim3((A.B + block - 1) / block), dim3(block), 0, stream, A,
                                   A.sd_end, A.sd_end_hi ? A.sd_end_hi : A.sd_end);
            else
                hipLaunchKernelGGL(tpr::lane_solve_kernel, dim3((A.B + block - 1) / block), dim3(block), 0, stream, A);
            return TPR_E_OK;
        }
        default:
            return fail(TPR_E_UNSUPPORTED, "unknown kernel variant");
    }
}

// Host-buffer calls on a handful of trajectories (the reference's own use: ONE trajectory per
// compute_parameterization call) are all latency.  Instead of a dozen stream-ordered allocations and pageable copies
// they go through one page-locked, device-mapped arena per calling thread: the inputs are packed into it by the CPU,
// the kernel reads them over PCIe (every input is fetched once, by coalesced loads issued together) and writes its
// outputs straight back into it; one launch, one stream synchronisation, two memcpy's on the host.
constexpr int kNotSmall = 1;
constexpr size_t kSmallCallBytes = 1 << 20;
struct HostArena {
    void *ptr = nullptr;
    size_t cap = 0;
    int device = -1;
};
thread_local HostArena g_arena;  // (never freed: a thread's exit may come after the runtime's own teardown)

int solve_small_host_call(const tpr_problem *p, const tpr_result *r, hipStream_t stream, int device) {
    const size_t B = (size_t)p->B, d = (size_t)p->d, nseg = (size_t)p->nseg, N = (size_t)p->N;
    struct Piece { const void *src; void *dst; size_t bytes, off; };
    Piece in[7] = {{p->coef, nullptr, B * 4 * nseg * d * 8, 0},
                   {p->breaks, nullptr, ((p->flags & TPR_BREAKS_PER_TRAJ) ? B : 1) * (nseg + 1) * 8, 0},
                   {p->grid, nullptr, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1) * 8, 0},
                   {p->vlim, nullptr, B * d * 16, 0}, {p->alim, nullptr, B * d * 16, 0},
                   {p->sd_start, nullptr, B * 8, 0}, {p->sd_end, nullptr, B * 8, 0}};
    Piece out[6] = {{nullptr, r->sd2, B * (N + 1) * 8, 0}, {nullptr, r->sd, B * (N + 1) * 8, 0}, {nullptr, r->u, B * N * 8, 0},
                    {nullptr, r->K, B * (N + 1) * 16, 0}, {nullptr, r->status, B * 4, 0},
                    {p->active, p->active, B * 16, 0}};  // (in and out)
    size_t total = 0;
    for (auto &q : in) { q.off = total; if (q.src) total += (q.bytes + 15) & ~(size_t)15; }
    for (auto &q : out) { q.off = total; if (q.dst) total += (q.bytes + 15) & ~(size_t)15; }
    if (total > kSmallCallBytes) return kNotSmall;
    tpr::BatchArgs A{};
    A.B = p->B; A.d = p->d; A.nseg = p->nseg; A.N = p->N; A.flags = p->flags;
    A.sd2 = r->sd2; A.sd = r->sd; A.u = r->u; A.K = r->K; A.status = r->status;  // (what is asked for decides the family)
    A.active = p->active;
    if (pick_variant(p->variant, A) != 4) return kNotSmall;  // the other families want workspaces: the general path
    if (g_arena.cap < total || g_arena.device != device) {
        if (g_arena.ptr) (void)hipHostFree(g_arena.ptr);
        g_arena = HostArena{};
        void *mem = nullptr;
        const size_t cap = total > (size_t)(256 << 10) ? kSmallCallBytes : (size_t)(256 << 10);
        if (hipHostMalloc(&mem, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return kNotSmall; }
        g_arena.ptr = mem; g_arena.cap = cap; g_arena.device = device;
    }
    char *base = static_cast<char *>(g_arena.ptr);
    for (auto &q : in) if (q.src) std::memcpy(base + q.off, q.src, q.bytes);
    if (out[5].src) std::memcpy(base + out[5].off, out[5].src, out[5].bytes);
    auto at = [&](const Piece &q) { return reinterpret_cast<double *>(base + q.off); };
    A.coef = at(in[0]); A.breaks = at(in[1]); A.grid = at(in[2]);
    A.vlim = in[3].src ? at(in[3]) : nullptr; A.alim = in[4].src ? at(in[4]) : nullptr;
    A.sd_start = in[5].src ? at(in[5]) : nullptr; A.sd_end = in[6].src ? at(in[6]) : nullptr;
    A.sd2 = out[0].dst ? at(out[0]) : nullptr; A.sd = out[1].dst ? at(out[1]) : nullptr;
    A.u = out[2].dst ? at(out[2]) : nullptr; A.K = out[3].dst ? at(out[3]) : nullptr;
    A.status = out[4].dst ? reinterpret_cast<int32_t *>(base + out[4].off) : nullptr;
    A.active = out[5].dst ? reinterpret_cast<int32_t *>(base + out[5].off) : nullptr;
    if (int rc = launch_solve(p, A, stream)) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    for (auto &q : out) if (q.dst) std::memcpy(q.dst, base + q.off, q.bytes);
    return TPR_E_OK;
}

}  // namespace

extern "C" {

const char *tpr_last_error(void) { return g_err.c_str(); }

const char *tpr_version(void) { return "toppra_hip 0.2 (gfx950)"; }
// ABI guard: the structures of this header grow at the END from version to version (0.2: tpr_problem.active); a binding built
// against an older header would pass a shorter structure, so it compares these sizes with its own before the first call.
int tpr_abi_sizes(int32_t *problem_bytes, int32_t *result_bytes, int32_t *dense_problem_bytes) {
    if (problem_bytes) *problem_bytes = (int32_t)sizeof(tpr_problem);
    if (result_bytes) *result_bytes = (int32_t)sizeof(tpr_result);
    if (dense_problem_bytes) *dense_problem_bytes = (int32_t)sizeof(tpr_dense_problem);
    return 2;  // ABI revision
}

int tpr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tpr_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(TPR_E_HIP, "no HIP device visible");
    if (device < 0 || device >= n || device >= 64) return fail(TPR_E_BADARG, "device index out of range");
    if (!g_checked[device].load(std::memory_order_acquire)) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(TPR_E_UNSUPPORTED, std::string("this library is built for gfx950 only, found ") + prop.gcnArchName);
        // Workspaces and host-call staging come from the device's stream-ordered pool (hipMallocAsync).  Its default
        // release threshold is 0: whatever was freed goes back to the driver at the next synchronisation, and a call that
        // needs a 1.2 GB workspace (tpr_param_spline_batch at the headline shape) maps it afresh every time -- 65 ms per
        // call instead of 2 on some boxes (profiles/r03: the kernel itself takes 1.9 ms).  Keep freed memory in the pool.
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
        g_checked[device].store(true, std::memory_order_release);
    }
    t_device = device;  // the caller's current device is left alone: every entry scopes its own (DeviceScope)
    g_process_device.store(device, std::memory_order_relaxed);
    return TPR_E_OK;
}

#ifdef TPR_DEBUG_PREDICT  // debug builds only: read and clear the walk's give-up counters
int tpr_debug_walk_fail(unsigned long long *out16) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(tpr::g_walk_fail), 16 * sizeof(unsigned long long)));
    unsigned long long zero[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(tpr::g_walk_fail), zero, sizeof(zero)));
    return TPR_E_OK;
}
int tpr_debug_walk_hist(unsigned int *out4x512) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out4x512, HIP_SYMBOL(tpr::g_walk_hist), 5 * 512 * sizeof(unsigned int)));
    return TPR_E_OK;
}
#endif

int tpr_solve_batch(const tpr_problem *p, const tpr_result *r, void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!r) return fail(TPR_E_BADARG, "null result");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    if (!(p->flags & TPR_DEVICE_PTRS) && p->B > 0) {
        const int rc = solve_small_host_call(p, r, stream, scope.dev);
        if (rc != kNotSmall) return rc;
    }
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    A.active = S.out(p->active, B * 4, true);  // the wrapper object's warm-start state, in / out
    A.sd2 = S.out(r->sd2, B * (N + 1));
    A.sd = S.out(r->sd, B * (N + 1));
    A.status = S.out(r->status, B);
    // outputs the caller does not want (the controllable sets, which the forward scan still reads; the path
    // accelerations, which retiming -- compute_trajectory -- never looks at) live in a stream-ordered workspace:
    // no host copy, no caller buffer, and the fast kernel family still serves the call
    auto workspace = [&](size_t doubles) -> double * {
        void *ws = nullptr;
        if (doubles == 0) return nullptr;
        if (S.err == hipSuccess) S.err = hipMallocAsync(&ws, doubles * sizeof(double), stream);
        if (S.err == hipSuccess) S.owned.push_back(ws);
        return static_cast<double *>(ws);
    };
    // (family 4 keeps K in LDS and skips what is not asked for: no workspace, no HBM traffic for it)
    A.u = r->u ? S.out(r->u, B * N) : nullptr;
    A.K = r->K ? S.out(r->K, B * (N + 1) * 2) : nullptr;
    {
        tpr::BatchArgs probe = A;  // the variant the call will take, with every output it could ask a workspace for
        if (!probe.u) probe.u = reinterpret_cast<double *>(8);
        if (!probe.K) probe.K = reinterpret_cast<double *>(8);
        const int v = pick_variant(p->variant, probe);
        if (v != 4 && v != 5) {  // (families 4 and 5 keep K in LDS and skip what is not asked for)
            if (!A.u) A.u = workspace(B * N);
            if (!A.K) A.K = workspace(B * (N + 1) * 2);
        }
    }
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (int rc = launch_solve(p, A, stream)) return rc;
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_solve_desired_duration_batch(const tpr_problem *p, const double *desired, double atol,
                                     const tpr_result *r, double *alpha, void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!r || !r->K || !desired) return fail(TPR_E_BADARG, "result.K and desired are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    if (!group_supported(A)) return fail(TPR_E_UNSUPPORTED, "TOPPRAsd: dof out of range");
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    const double *ddes = S.in(desired, B);
    A.sd2 = S.out(r->sd2, B * (N + 1));
    A.sd = S.out(r->sd, B * (N + 1));
    A.u = S.out(r->u, B * N);
    A.K = S.out(r->K, B * (N + 1) * 2);
    A.status = S.out(r->status, B);
    double *dalpha = S.out(alpha, B);
    // workspace: fastest / slowest profiles, the bisection's worklist, a status / alpha array when the caller wants none
    double *ws = nullptr;
    int32_t *wstatus = nullptr, *wlist = nullptr;
    const size_t per = 2 * (N + 1) + 2 * N;
    // (every allocation joins S.owned as soon as it exists: an early return must not leak the ones before it)
    if (S.err == hipSuccess) { S.err = hipMallocAsync(reinterpret_cast<void **>(&ws), B * (per + 3) * sizeof(double) + 8, stream); if (S.err == hipSuccess) S.owned.push_back(ws); }
    if (S.err == hipSuccess) { S.err = hipMallocAsync(reinterpret_cast<void **>(&wlist), (B + 2) * sizeof(int32_t), stream); if (S.err == hipSuccess) S.owned.push_back(wlist); }
    if (S.err == hipSuccess && !A.status) {
        S.err = hipMallocAsync(reinterpret_cast<void **>(&wstatus), B * sizeof(int32_t) + 4, stream);
        if (S.err == hipSuccess) S.owned.push_back(wstatus);
        A.status = wstatus;
    }
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (!dalpha) dalpha = ws + B * per;
    if (A.B > 0) {
        double *xf = ws, *uf = ws + B * (N + 1), *xl = ws + B * (2 * N + 1), *ul = ws + B * (3 * N + 2);
        double *wdur = ws + B * (per + 1);  // [B][2]: the two profiles' durations, summed by family 3's forward scans
        const double *dur_in = nullptr;
        tpr::BatchArgs Ab = A;
        Ab.backward_only = 1;
        // p->variant: 0 = auto; 2 / 3 force the rows-across-lanes scans / the certified lane kernel for both scans
        const bool fused = p->variant == 3 || (p->variant == 0 && cert_supported(Ab) && A.B >= cert_auto_from(A.d));
        if (fused) {
            // family 3: backward scan + fastest / slowest forward profiles in ONE launch (cert_solve_kernel<SDFWD>)
            if (!cert_supported(Ab)) return fail(TPR_E_UNSUPPORTED, "variant 3 needs an acceleration constraint, d <= 15, no strict mode");
            if (int rc = launch_cert_sd(A, xf, uf, xl, ul, wdur, stream)) return rc;
            dur_in = wdur;
        } else {
            // backward scan -> K and the controllability verdict (the time-optimal forward scan is not needed), then
            // the two forward scans on the rows-across-lanes kernel
            if (int rc = launch_solve(p, Ab, stream)) return rc;
            tpr::SdArgs F{A.B, A.nseg, A.N, A.flags, A.coef, A.breaks, A.grid, A.vlim, A.alim, A.sd_start, A.K,
                          A.status, xf, uf, xl, ul};
            if (int rc = dispatch_sd_forward(A.d, F, stream)) return rc;
        }
        tpr::SdBlendArgs G{A.B, A.N, A.flags, atol, A.grid, ddes, xf, uf, xl, ul, A.status,
                           A.sd2, A.sd, A.u, dalpha, A.status};
        const size_t finish_lds = 5 * (N + 1) * sizeof(double);
        if (finish_lds <= kMaxDynamicLds) G.dur = dur_in;  // (the three-kernel path below computes its own)
        if (finish_lds <= kMaxDynamicLds) {
            // one wave per trajectory: durations, bisection and the blend from LDS-resident profiles
            hipLaunchKernelGGL(tpr::sd_finish_kernel, dim3(A.B), dim3(64), finish_lds, stream, G);
        } else {
            HIP_TRY(hipMemsetAsync(wlist + B, 0, sizeof(int32_t), stream));  // the worklist's counter sits behind it
            hipLaunchKernelGGL(tpr::sd_decide_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, G, wlist, wlist + B);
            hipLaunchKernelGGL(tpr::sd_bisect_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, G, wlist, wlist + B);
            hipLaunchKernelGGL(tpr::sd_blend_kernel, dim3(A.B), dim3(64), 0, stream, G);
        }
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_robust_solve_batch(const tpr_problem *p, const double *ellipsoid, const tpr_result *r, double *X,
                           void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!r || !r->K || !ellipsoid) return fail(TPR_E_BADARG, "result.K and ellipsoid are required");
    if (!(p->flags & TPR_HAS_ACCELERATION)) return fail(TPR_E_BADARG, "the robust path needs an acceleration constraint");
    if (ellipsoid[0] < 0 || ellipsoid[1] < 0 || ellipsoid[2] < 0) return fail(TPR_E_BADARG, "ellipsoid axes must be non-negative");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::RobustArgs P{};
    P.A = stage_problem(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    P.A.sd2 = S.out(r->sd2, B * (N + 1));
    P.A.sd = S.out(r->sd, B * (N + 1));
    P.A.u = S.out(r->u, B * N);
    P.A.K = S.out(r->K, B * (N + 1) * 2);
    P.A.status = S.out(r->status, B);
    P.X = S.out(X, B * (N + 1) * 2);
    P.ru = ellipsoid[0]; P.rx = ellipsoid[1]; P.rc = ellipsoid[2];
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (P.A.B > 0) {
        // p->variant: 0 = auto, 1 = the generic lane kernel (one trajectory per lane, rows in scratch), 2 = rows across lanes
        if (p->variant != 1 && group_supported(P.A)) {  // up to 16 dof, Interpolation or Collocation, with or without feasible sets
            if (int rc = dispatch_group_robust(P, stream)) return rc;
        } else {
            (void)tpr_tu_robust_lane_launch(&P, stream);
        }
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_solve_batch_timed(const tpr_problem *p, const tpr_result *r, void *stream_, int reps,
                          float *ms_per_launch) {
    if (int rc = check_problem(p)) return rc;
    if (!(p->flags & TPR_DEVICE_PTRS)) return fail(TPR_E_BADARG, "timed entry needs TPR_DEVICE_PTRS");
    if (!r || !r->K || reps < 1 || !ms_per_launch) return fail(TPR_E_BADARG, "bad timed arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(true, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    A.sd2 = r->sd2; A.sd = r->sd; A.u = r->u; A.K = r->K; A.status = r->status;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, stream));
    for (int i = 0; i < reps; ++i)
        if (int rc = launch_solve(p, A, stream)) return rc;
    HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(hipGetLastError());
    *ms_per_launch = ms / (float)reps;
    return TPR_E_OK;
}

int tpr_controllable_sets_batch(const tpr_problem *p, const double *sdmin, const double *sdmax,
                                double *K, void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!sdmin || !sdmax || !K) return fail(TPR_E_BADARG, "sdmin/sdmax/K are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    const double *dmin = S.in(sdmin, B), *dmax = S.in(sdmax, B);
    A.K = S.out(K, B * (N + 1) * 2);
    A.active = S.out(p->active, B * 4, true);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0) {
        if ((group_supported(A) || wave_supported(A)) && A.N >= 1) {  // the backward scan of the fast kernels
            A.sd_end = dmin;
            A.sd_end_hi = dmax;
            A.backward_only = 1;
            if (int rc = launch_solve(p, A, stream)) return rc;
        } else {
            hipLaunchKernelGGL(tpr::lane_controllable_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A,
                               dmin, dmax);
        }
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

// ---- dense rows (any canonical-linear constraint list): tpr_dense.hip.inc ----------------------------------------
namespace tpr {
// compute_reachable_sets (reachability_algorithm.py:378-431) on dense rows: lane_reachable_kernel (tpr_lane.hip.inc: one
// trajectory per lane, the reference's solve_stagewise_optim with its stateful warm start, the deltas[i - 1] quirk of
// _one_step_forward) with the stage rows copied from the arrays instead of generated.
static __global__ void __launch_bounds__(64) lane_dense_reachable_kernel(DenseArgs A, const double *sdmin, const double *sdmax,
                                                                  double *L, double *X) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    const int N = A.N, nC = A.nC;
    const double *ga = A.a + (size_t)b * (N + 1) * nC, *gb = A.b + (size_t)b * (N + 1) * nC, *gc = A.c + (size_t)b * (N + 1) * nC;
    const double *glow = A.low + (size_t)b * 2 * (N + 1), *ghigh = A.high + (size_t)b * 2 * (N + 1);
    const double *deltas = A.deltas + (size_t)b * N;
    StageRows R;
    WarmStart W = {{0, 0}, {0, 0}};
    if (A.active) { const int32_t *st = A.active + (size_t)b * 4; W.up[0] = st[0]; W.up[1] = st[1]; W.down[0] = st[2]; W.down[1] = st[3]; }
    auto put_state = [&]() {
        if (A.active) { int32_t *st = A.active + (size_t)b * 4; st[0] = W.up[0]; st[1] = W.up[1]; st[2] = W.down[0]; st[3] = W.down[1]; }
    };
    unsigned char order[kMaxRows];
    auto rows = [&](int i) {
        R.nC = nC;
        for (int r = 2; r < nC; ++r) { R.a[r] = ga[(size_t)i * nC + r]; R.b[r] = gb[(size_t)i * nC + r]; R.c[r] = gc[(size_t)i * nC + r]; }
        R.low0 = glow[2 * i]; R.high0 = ghigh[2 * i]; R.low1 = glow[2 * i + 1]; R.high1 = ghigh[2 * i + 1];
    };
    double *Xb = X + (size_t)b * 2 * (N + 1), *Lb = L + (size_t)b * 2 * (N + 1);
    for (int i = 0; i <= N; ++i) {  // feasible sets (:131-164), on the same wrapper object
        const bool last = i == N;
        rows(i);
        set_next_rows(R, last, last ? 0.0 : deltas[i], -kFeasMaxX, kFeasMaxX);
        double uu, lo, hi;
        stage_solve(R, W, 1e-9, 1.0, -kFeasMaxX, kFeasMaxX, 1, order, uu, lo);
        stage_solve(R, W, -1e-9, -1.0, -kFeasMaxX, kFeasMaxX, 1, order, uu, hi);
        if (lo < 0) lo = 0;
        Xb[2 * i] = lo; Xb[2 * i + 1] = hi;
    }
    for (int i = 0; i <= N; ++i) { Lb[2 * i] = 0.0; Lb[2 * i + 1] = 0.0; }
    double l0 = boundary_x(A.flags, sdmin[b]), l1 = boundary_x(A.flags, sdmax[b]);
    Lb[0] = l0; Lb[1] = l1;
    for (int i = 0; i < N; ++i) {
        const double delta = deltas[i];
        const double dprev = i > 0 ? deltas[i - 1] : deltas[N - 1];  // get_deltas()[i - 1]: Python's negative index at i = 0
        double lo, hi;
        if (isnan(l0) || isnan(l1)) { lo = qnan(); hi = qnan(); }
        else {
            rows(i);
            set_next_rows(R, false, delta, Xb[2 * (i + 1)], Xb[2 * (i + 1) + 1]);
            double uu, xx;
            stage_solve(R, W, -2 * dprev, -1.0, l0, l1, 1, order, uu, xx);
            hi = xx + 2 * dprev * uu;
            stage_solve(R, W, 2 * dprev, 1.0, l0, l1, 1, order, uu, xx);
            lo = xx + 2 * dprev * uu;
        }
        if (lo < 0) lo = 0;
        Lb[2 * (i + 1)] = lo; Lb[2 * (i + 1) + 1] = hi;
        if (isnan(lo) || isnan(hi)) break;
        l0 = lo; l1 = hi;
    }
    put_state();
}
}  // namespace tpr

namespace {
int check_dense(const tpr_dense_problem *p) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p) return fail(TPR_E_BADARG, "null dense problem");
    if (p->B < 0 || p->N < 1) return fail(TPR_E_BADARG, "dense problem: B >= 0, N >= 1");
    if (p->nC < 2 || p->nC > 122) return fail(TPR_E_UNSUPPORTED, "dense problem: 2 <= nC <= 122 rows per stage (incl. the two x_next rows)");
    if (!p->a || !p->b || !p->c || !p->low || !p->high || !p->deltas) return fail(TPR_E_BADARG, "dense problem: a, b, c, low, high, deltas are required");
    return TPR_E_OK;
}
tpr::DenseArgs stage_dense(const tpr_dense_problem *p, Staging &S) {
    tpr::DenseArgs A{};
    const size_t B = (size_t)p->B, N = (size_t)p->N, nC = (size_t)p->nC;
    A.B = p->B; A.N = p->N; A.nC = p->nC; A.flags = p->flags;
    A.a = S.in(p->a, B * (N + 1) * nC); A.b = S.in(p->b, B * (N + 1) * nC); A.c = S.in(p->c, B * (N + 1) * nC);
    A.low = S.in(p->low, B * (N + 1) * 2); A.high = S.in(p->high, B * (N + 1) * 2);
    A.deltas = S.in(p->deltas, B * N);
    A.active = S.out(p->active, B * 4, true);
    return A;
}
}  // namespace

int tpr_solve_dense_batch(const tpr_dense_problem *p, const tpr_result *r, void *stream_) {
    if (int rc = check_dense(p)) return rc;
    if (!r || !r->K) return fail(TPR_E_BADARG, "dense solve: r->K is required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->a));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::DenseArgs A = stage_dense(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    A.sd_start = S.in(p->sd_start, B); A.sd_end = S.in(p->sd_end, B);
    A.sd2 = S.out(r->sd2, B * (N + 1)); A.sd = S.out(r->sd, B * (N + 1)); A.u = S.out(r->u, B * N);
    A.K = S.out(r->K, B * (N + 1) * 2); A.status = S.out(r->status, B);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0 && tpr_tu_dense_launch(&A, 0, stream) != 0) return fail(TPR_E_UNSUPPORTED, "dense solve: no kernel for this row count");
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_solve_desired_duration_dense_batch(const tpr_dense_problem *p, const double *desired, double atol, const tpr_result *r,
                                           double *alpha, void *stream_) {
    if (int rc = check_dense(p)) return rc;
    if (!r || !r->K || !desired) return fail(TPR_E_BADARG, "result.K and desired are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->a));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::DenseArgs A = stage_dense(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    const double *ddes = S.in(desired, B);
    A.sd_start = S.in(p->sd_start, B); A.sd_end = S.in(p->sd_end, B);
    double *dsd2 = S.out(r->sd2, B * (N + 1)), *dsd = S.out(r->sd, B * (N + 1)), *du = S.out(r->u, B * N);
    A.K = S.out(r->K, B * (N + 1) * 2); A.status = S.out(r->status, B);
    double *dalpha = S.out(alpha, B);
    // workspace: fastest / slowest profiles, the bisection's worklist, a status / alpha array when the caller wants none
    double *ws = nullptr;
    int32_t *wstatus = nullptr, *wlist = nullptr;
    const size_t per = 2 * (N + 1) + 2 * N;
    // (every allocation joins S.owned as soon as it exists: an early return must not leak the ones before it)
    if (S.err == hipSuccess) { S.err = hipMallocAsync(reinterpret_cast<void **>(&ws), B * (per + (dalpha ? 0 : 1)) * sizeof(double) + 8, stream); if (S.err == hipSuccess) S.owned.push_back(ws); }
    if (S.err == hipSuccess) { S.err = hipMallocAsync(reinterpret_cast<void **>(&wlist), (B + 2) * sizeof(int32_t), stream); if (S.err == hipSuccess) S.owned.push_back(wlist); }
    if (S.err == hipSuccess && !A.status) {
        S.err = hipMallocAsync(reinterpret_cast<void **>(&wstatus), B * sizeof(int32_t) + 4, stream);
        if (S.err == hipSuccess) S.owned.push_back(wstatus);
        A.status = wstatus;
    }
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (!dalpha) dalpha = ws + B * per;
    if (A.B > 0) {
        A.sd_xf = ws; A.sd_uf = ws + B * (N + 1); A.sd_xl = ws + B * (2 * N + 1); A.sd_ul = ws + B * (3 * N + 2);
        // backward scan -> K and the controllability verdict, the two forward scans, then durations / bisection / blend
        A.backward_only = 1;
        if (tpr_tu_dense_launch(&A, 0, stream) != 0 || tpr_tu_dense_launch(&A, 2, stream) != 0)
            return fail(TPR_E_UNSUPPORTED, "dense TOPPRAsd: no kernel for this row count");
        tpr::SdBlendArgs G{A.B, A.N, A.flags, atol, nullptr, ddes, A.sd_xf, A.sd_uf, A.sd_xl, A.sd_ul, A.status,
                           dsd2, dsd, du, dalpha, A.status, A.deltas};
        const size_t finish_lds = 5 * (N + 1) * sizeof(double);
        if (finish_lds <= kMaxDynamicLds) {
            hipLaunchKernelGGL(tpr::sd_finish_kernel, dim3(A.B), dim3(64), finish_lds, stream, G);
        } else {
            HIP_TRY(hipMemsetAsync(wlist + B, 0, sizeof(int32_t), stream));
            hipLaunchKernelGGL(tpr::sd_decide_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, G, wlist, wlist + B);
            hipLaunchKernelGGL(tpr::sd_bisect_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, G, wlist, wlist + B);
            hipLaunchKernelGGL(tpr::sd_blend_kernel, dim3(A.B), dim3(64), 0, stream, G);
        }
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_reachable_sets_dense_batch(const tpr_dense_problem *p, const double *sdmin, const double *sdmax, double *L, double *X,
                                   void *stream_) {
    if (int rc = check_dense(p)) return rc;
    if (!sdmin || !sdmax || !L) return fail(TPR_E_BADARG, "sdmin/sdmax/L are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->a));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::DenseArgs A = stage_dense(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    const double *dmin = S.in(sdmin, B), *dmax = S.in(sdmax, B);
    double *dL = S.out(L, B * (N + 1) * 2);
    double *dX = S.out(X, B * (N + 1) * 2);
    if (!dX && B > 0) {  // the feasible sets are an intermediate when the caller does not ask for them
        void *ws = nullptr;
        if (S.err == hipSuccess) S.err = hipMallocAsync(&ws, B * (N + 1) * 2 * sizeof(double), stream);
        if (S.err == hipSuccess) S.owned.push_back(ws);
        dX = static_cast<double *>(ws);
    }
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0)
        hipLaunchKernelGGL(tpr::lane_dense_reachable_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A, dmin, dmax, dL, dX);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_controllable_sets_dense_batch(const tpr_dense_problem *p, const double *sdmin, const double *sdmax, double *K,
                                      void *stream_) {
    if (int rc = check_dense(p)) return rc;
    if (!sdmin || !sdmax || !K) return fail(TPR_E_BADARG, "sdmin/sdmax/K are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->a));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::DenseArgs A = stage_dense(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    A.sd_end = S.in(sdmin, B); A.sd_end_hi = S.in(sdmax, B);
    A.K = S.out(K, B * (N + 1) * 2);
    A.backward_only = 1;
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0 && tpr_tu_dense_launch(&A, 0, stream) != 0) return fail(TPR_E_UNSUPPORTED, "dense controllable sets: no kernel for this row count");
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_feasible_sets_dense_batch(const tpr_dense_problem *p, double *X, void *stream_) {
    if (int rc = check_dense(p)) return rc;
    if (!X) return fail(TPR_E_BADARG, "X is required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->a));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::DenseArgs A = stage_dense(p, S);
    A.X = S.out(X, (size_t)p->B * ((size_t)p->N + 1) * 2);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0 && tpr_tu_dense_launch(&A, 1, stream) != 0) return fail(TPR_E_UNSUPPORTED, "dense feasible sets: no kernel for this row count");
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_reachable_sets_batch(const tpr_problem *p, const double *sdmin, const double *sdmax, double *L, double *X,
                             void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!sdmin || !sdmax || !L) return fail(TPR_E_BADARG, "sdmin/sdmax/L are required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    const double *dmin = S.in(sdmin, B), *dmax = S.in(sdmax, B);
    double *dL = S.out(L, B * (N + 1) * 2);
    double *dX = S.out(X, B * (N + 1) * 2);
    if (!dX && B > 0) {  // the feasible sets are an intermediate when the caller does not ask for them
        void *ws = nullptr;
        if (S.err == hipSuccess) S.err = hipMallocAsync(&ws, B * (N + 1) * 2 * sizeof(double), stream);
        if (S.err == hipSuccess) S.owned.push_back(ws);
        dX = static_cast<double *>(ws);
    }
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0)
        hipLaunchKernelGGL(tpr::lane_reachable_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A, dmin, dmax, dL, dX);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_feasible_sets_batch(const tpr_problem *p, double *X, void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!X) return fail(TPR_E_BADARG, "X is required");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    double *dX = S.out(X, (size_t)p->B * (p->N + 1) * 2);
    A.active = S.out(p->active, (size_t)p->B * 4, true);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0) {
        // p->variant: 0 = auto, 2 / 3 / 4 force a kernel family (1: the generic lane kernel)
        const int want = p->variant;
        const bool wave_auto = wave_supported(A) && (A.active || A.B <= 64 || !group_supported(A));
        if (want == 3 && !cert_feasible_supported(A))
            return fail(TPR_E_UNSUPPORTED, "variant 3 needs an acceleration constraint, d <= 15, no strict mode, no warm-start state");
        if (want == 4 && !wave_supported(A)) return fail(TPR_E_UNSUPPORTED, "variant 4: N too large for the per-trajectory LDS tables");
        if (A.active && (want == 2 || want == 3))
            return fail(TPR_E_UNSUPPORTED, "tpr_problem.active (warm-start state in / out) is maintained by kernel families 4 and 1 only: leave variant at 0");
        if (A.active && want == 0 && !wave_supported(A)) {  // N > 1480: the generic lane kernel carries the state
            hipLaunchKernelGGL(tpr::lane_feasible_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A, dX);
        } else if (want == 4 || (want == 0 && wave_auto)) {
            // one trajectory per wave: a handful of trajectories (latency), 17..32 dof, or the wrapper object's
            // warm-start state in / out
            A.feasible_X = dX;
            if (int rc = launch_wave(A, stream)) return rc;
        } else if (want == 3 || (want == 0 && cert_feasible_supported(A) && A.B >= (A.d <= 8 ? 8192 : cert_auto_from(A.d)))) {
            // one trajectory per lane, certified answers (family 3): a fixed-latency round up to 65536 trajectories
            if (int rc = launch_cert_feasible(A, dX, stream)) return rc;
        } else if (group_supported(A) && want != 1) {
            if (int rc = dispatch_group_feasible(A, dX, stream)) return rc;
        } else {
            hipLaunchKernelGGL(tpr::lane_feasible_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A, dX);
        }
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_constraint_params_batch(const tpr_problem *p, double *a, double *b, double *c, double *low,
                                double *high, double *xbound, double *qs, double *qss, void *stream_) {
    if (int rc = check_problem(p)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    const size_t pts = (size_t)p->B * (p->N + 1), nC = (size_t)rows_per_lp(p);
    double *da = S.out(a, pts * nC), *db = S.out(b, pts * nC), *dc = S.out(c, pts * nC);
    double *dlow = S.out(low, pts * 2), *dhigh = S.out(high, pts * 2), *dxb = S.out(xbound, pts * 2);
    double *dqs = S.out(qs, pts * p->d), *dqss = S.out(qss, pts * p->d);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (pts > 0) {
        const unsigned tiles = (unsigned)((p->N + 1 + tpr::kParamsTile - 1) / tpr::kParamsTile);
        if (tiles > 65535u) return fail(TPR_E_UNSUPPORTED, "tpr_constraint_params_batch: more than 2 M gridpoints");
        const int tile = (int)((p->N + 1 + tiles - 1) / tiles);
        const size_t lds = ((size_t)2 * (tpr::kParamsTile + 1) * p->d + tpr::kParamsTile + 2 * p->d) * sizeof(double);
        hipLaunchKernelGGL(tpr::params_tile_kernel, dim3((unsigned)p->B, tiles), dim3(256), lds, stream, A, tile,
                           da, db, dc, dlow, dhigh, dxb, dqs, dqss);
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_solve_stagewise_batch(const tpr_problem *p, const int32_t *stage, const double *g,
                              const double *xb, int32_t *active, int solve_lp1d, double *out,
                              void *stream_) {
    if (int rc = check_problem(p)) return rc;
    if (!stage || !g || !xb || !active || !out) return fail(TPR_E_BADARG, "null stagewise argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    tpr::BatchArgs A = stage_problem(p, S);
    const size_t B = (size_t)p->B;
    const int32_t *dstage = S.in(stage, B);
    const double *dg = S.in(g, B * 2), *dxb = S.in(xb, B * 4);
    int32_t *dact = S.out(active, B * 4, true);
    double *dout = S.out(out, B * 2);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0)
        hipLaunchKernelGGL(tpr::lane_stagewise_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A, dstage,
                           dg, dxb, dact, solve_lp1d, dout);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_const_accel_times_batch(const tpr_problem *p, const double *sd, double *ts, double *us, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p || p->B < 0 || p->N < 1 || !p->grid || !sd || !ts) return fail(TPR_E_BADARG, "bad const-accel arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->grid));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    const size_t B = (size_t)p->B, N = (size_t)p->N;
    tpr::ParamArgs A{};
    A.B = p->B; A.N = p->N; A.flags = p->flags;
    A.grid = S.in(p->grid, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1));
    A.sd = S.in(sd, B * (N + 1));
    A.ts = S.out(ts, B * (N + 1));
    A.us = S.out(us, B * N);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (A.B > 0) {
        const size_t lds = 2 * ((size_t)A.N + 1) * sizeof(double);
        if (lds <= kMaxDynamicLds) hipLaunchKernelGGL(tpr::const_accel_times_kernel, dim3(A.B), dim3(64), lds, stream, A);
        else hipLaunchKernelGGL(tpr::const_accel_times_lane_kernel, dim3((A.B + 63) / 64), dim3(64), 0, stream, A);
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_const_accel_eval_batch(const tpr_problem *p, const double *sd, const double *ts, const double *us,
                               int T, const double *times, int order, double *out, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p || p->B < 0 || p->N < 1 || p->d < 1 || p->nseg < 1 || !p->coef || !p->breaks || !p->grid || !sd || !ts ||
        !us || !times || !out || T < 0 || order < 0 || order > 2)
        return fail(TPR_E_BADARG, "bad const-accel eval arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    const size_t B = (size_t)p->B, N = (size_t)p->N, d = (size_t)p->d, nseg = (size_t)p->nseg;
    tpr::EvalArgs A{};
    A.B = p->B; A.N = p->N; A.d = p->d; A.nseg = p->nseg; A.T = T; A.flags = p->flags; A.order = order;
    A.coef = S.in(p->coef, B * 4 * nseg * d);
    A.breaks = S.in(p->breaks, ((p->flags & TPR_BREAKS_PER_TRAJ) ? B : 1) * (nseg + 1));
    A.grid = S.in(p->grid, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1));
    A.sd = S.in(sd, B * (N + 1));
    A.ts = S.in(ts, B * (N + 1));
    A.us = S.in(us, B * N);
    A.times = S.in(times, B * (size_t)T);
    A.out = S.out(out, B * (size_t)T * d);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    const long long total = (long long)p->B * T * p->d;  // one thread per (sample, dof)
    if (total > (long long)0x7fffffff * 256) return fail(TPR_E_BADARG, "constant-acceleration evaluation: B T d too large for one launch");
    if (total > 0)
        hipLaunchKernelGGL(tpr::const_accel_eval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

namespace {
// Launch the fit of B*d splines of up to m points: short splines keep their working arrays in registers /
// scratch; longer ones use a stream-ordered global workspace of 6 m doubles per spline.
int launch_spline_fit(const tpr::SplineArgs &A, const int32_t *counts, Staging &S, hipStream_t stream) {
    const long long total = (long long)A.B * A.d;
    if (total <= 0) return TPR_E_OK;
    const dim3 grid((unsigned)((total + 127) / 128)), block(128);
    if (A.m <= 8) {  // the usual handful of waypoints: unrolled, arrays in registers
        hipLaunchKernelGGL((tpr::spline_fit_kernel<8>), grid, block, 0, stream, A, counts, (double *)nullptr);
        return TPR_E_OK;
    }
    if (A.m <= tpr::kSplineMaxPts) {
        hipLaunchKernelGGL((tpr::spline_fit_kernel<tpr::kSplineMaxPts>), grid, block, 0, stream, A, counts, (double *)nullptr);
        return TPR_E_OK;
    }
    void *ws = nullptr;
    hipError_t e = hipMallocAsync(&ws, (size_t)6 * A.m * (size_t)total * sizeof(double), stream);
    if (e != hipSuccess) return fail(TPR_E_HIP, std::string("spline-fit workspace: ") + hipGetErrorString(e));
    S.owned.push_back(ws);
    hipLaunchKernelGGL((tpr::spline_fit_kernel<0>), grid, block, 0, stream, A, counts, static_cast<double *>(ws));
    return TPR_E_OK;
}
}  // namespace

int tpr_spline_fit_batch(int B, int m, int d, const double *knots, int knots_per_path,
                         const double *waypoints, int bc_start, int bc_end, const double *bc_start_val,
                         const double *bc_end_val, double *coef, int device_ptrs, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (B < 0 || d < 1 || m < 2 || !knots || !waypoints || !coef)
        return fail(TPR_E_BADARG, "spline fit needs m >= 2 waypoints, knots, waypoints, coef");
    if (bc_start < 0 || bc_start > 2 || bc_end < 0 || bc_end > 2) return fail(TPR_E_BADARG, "unknown boundary condition");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(device_ptrs != 0, waypoints));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(device_ptrs != 0, stream);
    tpr::SplineArgs A{};
    A.B = B; A.m = m; A.d = d; A.knots_per_path = knots_per_path; A.bc0 = bc_start; A.bc1 = bc_end;
    A.knots = S.in(knots, (size_t)(knots_per_path ? B : 1) * m);
    A.way = S.in(waypoints, (size_t)B * m * d);
    A.bcv0 = S.in(bc_start_val, (size_t)B * d);
    A.bcv1 = S.in(bc_end_val, (size_t)B * d);
    A.coef = S.out(coef, (size_t)B * 4 * (m - 1) * d);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (int rc = launch_spline_fit(A, nullptr, S, stream)) return rc;
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_param_spline_batch(const tpr_problem *p, const double *sd, double *knot_times, int32_t *counts, double *coef_t,
                           void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p || p->B < 0 || p->N < 1 || p->d < 1 || p->nseg < 1 || !p->coef || !p->breaks || !p->grid || !sd || !knot_times ||
        !counts || !coef_t)
        return fail(TPR_E_BADARG, "bad spline-parametrizer arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    const size_t B = (size_t)p->B, N = (size_t)p->N, d = (size_t)p->d, nseg = (size_t)p->nseg;
    tpr::ParamSplineArgs K{};
    K.B = p->B; K.N = p->N; K.d = p->d; K.nseg = p->nseg; K.flags = p->flags;
    K.coef = S.in(p->coef, B * 4 * nseg * d);
    K.breaks = S.in(p->breaks, ((p->flags & TPR_BREAKS_PER_TRAJ) ? B : 1) * (nseg + 1));
    K.grid = S.in(p->grid, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1));
    K.sd = S.in(sd, B * (N + 1));
    K.tk = S.out(knot_times, B * (N + 1));
    K.counts = S.out(counts, B);
    double *dcoef = S.out(coef_t, B * 4 * N * d);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (p->variant < 0 || p->variant > 3) return fail(TPR_E_BADARG, "spline parametrizer: variant 0 (auto), 1 (generic), 2 (fused, LAPACK order), 3 (knot-parallel)");
    // knot-parallel kernel (tpr_spline.hip.inc): a block per trajectory, all N + 1 knots with their d right-hand sides in LDS
    const size_t pcr_lds = (1 + d + std::max<size_t>(d - 1, 2)) * ((N + 1) | 1) * sizeof(double);
    const bool pcr_fits = d <= 16 && N + 1 <= 1024 && pcr_lds <= kMaxDynamicLds - 256;
    if (p->variant == 3 && !pcr_fits)
        return fail(TPR_E_UNSUPPORTED, "spline parametrizer variant 3 needs d <= 16 and about 2 (d + 1) (N + 1) doubles of LDS (<= 64 KB)");
    if (B > 0 && pcr_fits && (p->variant == 0 || p->variant == 3)) {
        const int kpt = N + 1 <= 256 ? 1 : (N + 1 <= 512 ? 2 : 4);
        int pcr_debug = 0;
#if TPR_PS_EXPERIMENT
        if (const char *e = getenv("TPR_PS_DEBUG")) pcr_debug = atoi(e);
#endif
        const dim3 grid((unsigned)B), block(256);
#define TPR_PCR_CASE(DD) \
        case DD: \
            if (kpt == 1) hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 1>), grid, block, pcr_lds, stream, K, dcoef, pcr_debug); \
            else if (kpt == 2) hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 2>), grid, block, pcr_lds, stream, K, dcoef, pcr_debug); \
            else hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 4>), grid, block, pcr_lds, stream, K, dcoef, pcr_debug); \
            break
        switch (p->d) {
            TPR_PCR_CASE(1); TPR_PCR_CASE(2); TPR_PCR_CASE(3); TPR_PCR_CASE(4);
            TPR_PCR_CASE(5); TPR_PCR_CASE(6); TPR_PCR_CASE(7); TPR_PCR_CASE(8);
            TPR_PCR_CASE(9); TPR_PCR_CASE(10); TPR_PCR_CASE(11); TPR_PCR_CASE(12);
            TPR_PCR_CASE(13); TPR_PCR_CASE(14); TPR_PCR_CASE(15); TPR_PCR_CASE(16);
        }
#undef TPR_PCR_CASE
        HIP_TRY(S.finish());
        return TPR_E_OK;
    }
    if (B > 0 && d <= 64 && N <= 65535 && p->variant != 1) {
        // one kernel (tpr_spline.hip.inc): a wave owns floor(64/d) trajectories; workspace = the eliminated right-hand
        // sides [tasks][N+1][64], the eliminated matrix rows [tasks][N+1][3][tpw], the knots' path positions [B][N+1]
        tpr::ParamFusedArgs F{};
        F.K = K; F.coef_t = dcoef; F.tpw = std::min(64 / (int)d, tpr::kPsMaxTpw);
        const size_t tasks = (B + F.tpw - 1) / F.tpw;
        const size_t rhs_n = tasks * (N + 1) * 64, rows_n = tasks * (N + 1) * 3 * F.tpw, sk_n = B * (N + 1);
        void *ws = nullptr;
        HIP_TRY(hipMallocAsync(&ws, (rhs_n + rows_n + sk_n) * sizeof(double), stream));
        S.owned.push_back(ws);
        F.rhs = static_cast<double *>(ws);
        F.rows = F.rhs + rhs_n;
        F.sk = F.rows + rows_n;
        int tile = tpr::kPsTile;
#if TPR_PS_EXPERIMENT
        if (const char *e = getenv("TPR_PS_TILE")) tile = atoi(e);
        if (const char *e = getenv("TPR_PS_DEBUG")) F.debug = atoi(e);
#endif
        const size_t lds = ((size_t)(tile + 1) * (128 + 5 * F.tpw) + 64) * sizeof(double) + (size_t)F.tpw * sizeof(int);
        if (tile == 4) hipLaunchKernelGGL(tpr::param_spline_fused_kernel<4>, dim3((unsigned)tasks), dim3(64), lds, stream, F);
        else if (tile == 16) hipLaunchKernelGGL(tpr::param_spline_fused_kernel<16>, dim3((unsigned)tasks), dim3(64), lds, stream, F);
        else hipLaunchKernelGGL(tpr::param_spline_fused_kernel<tpr::kPsTile>, dim3((unsigned)tasks), dim3(64), lds, stream, F);
        HIP_TRY(S.finish());
        return TPR_E_OK;
    }
    // generic path (d > 64, or variant 1): waypoints q(s_i) and the two end derivatives, then the spline-fit kernel
    void *ws = nullptr;
    if (S.err == hipSuccess && B > 0) S.err = hipMallocAsync(&ws, (B * (N + 1) * d + 2 * B * d) * sizeof(double), stream);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (B > 0) {
        S.owned.push_back(ws);
        K.way = static_cast<double *>(ws);
        K.bcv0 = K.way + B * (N + 1) * d;
        K.bcv1 = K.bcv0 + B * d;
        hipLaunchKernelGGL(tpr::param_spline_knots_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, K);
        tpr::SplineArgs A{};
        A.B = p->B; A.m = p->N + 1; A.d = p->d; A.knots_per_path = 1; A.bc0 = tpr::kBcFirst; A.bc1 = tpr::kBcFirst;
        A.knots = K.tk; A.way = K.way; A.bcv0 = K.bcv0; A.bcv1 = K.bcv1; A.coef = dcoef;
        if (int rc = launch_spline_fit(A, K.counts, S, stream)) return rc;
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_param_spline_sample_batch(const tpr_problem *p, const double *sd, int T, const double *times, int times_per_traj,
                                  int fractions, double *q, double *qd, double *qdd, double *duration, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (!p || p->B < 0 || p->N < 1 || p->d < 1 || p->nseg < 1 || !p->coef || !p->breaks || !p->grid || !sd || !times || T < 0 ||
        (!q && !qd && !qdd))
        return fail(TPR_E_BADARG, "bad spline-sampling arguments");
    if (!times_per_traj && !fractions) return fail(TPR_E_BADARG, "shared sample times must be fractions of each trajectory's duration");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(p->flags & TPR_DEVICE_PTRS, p->coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(p->flags & TPR_DEVICE_PTRS, stream);
    const size_t B = (size_t)p->B, N = (size_t)p->N, d = (size_t)p->d, nseg = (size_t)p->nseg;
    tpr::ParamSplineArgs K{};
    K.B = p->B; K.N = p->N; K.d = p->d; K.nseg = p->nseg; K.flags = p->flags;
    K.coef = S.in(p->coef, B * 4 * nseg * d);
    K.breaks = S.in(p->breaks, ((p->flags & TPR_BREAKS_PER_TRAJ) ? B : 1) * (nseg + 1));
    K.grid = S.in(p->grid, ((p->flags & TPR_GRID_PER_TRAJ) ? B : 1) * (N + 1));
    K.sd = S.in(sd, B * (N + 1));
    tpr::ParamSampleArgs Q{};
    Q.T = T; Q.fractions = fractions; Q.times_per_traj = times_per_traj;
    Q.times = S.in(times, (times_per_traj ? B : 1) * (size_t)T);
    Q.q[0] = S.out(q, B * T * d); Q.q[1] = S.out(qd, B * T * d); Q.q[2] = S.out(qdd, B * T * d);
    Q.duration = S.out(duration, B);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    const size_t lds = (1 + d + std::max<size_t>(d, 2)) * ((N + 1) | 1) * sizeof(double);
    if (!(d <= 16 && N + 1 <= 1024 && lds <= kMaxDynamicLds - 256))
        return fail(TPR_E_UNSUPPORTED, "spline sampling: d <= 16 and about (2 d + 1) (N + 1) doubles of LDS (<= 64 KB); use tpr_param_spline_batch + tpr_ppoly_eval_batch");
    if (B > 0 && T > 0) {
        const int kpt = N + 1 <= 256 ? 1 : (N + 1 <= 512 ? 2 : 4);
        const dim3 grid((unsigned)B), block(256);
#define TPR_PCRS_CASE(DD) \
        case DD: \
            if (kpt == 1) hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 1, true>), grid, block, lds, stream, K, (double *)nullptr, 0, Q); \
            else if (kpt == 2) hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 2, true>), grid, block, lds, stream, K, (double *)nullptr, 0, Q); \
            else hipLaunchKernelGGL((tpr::param_spline_pcr_kernel<DD, 4, true>), grid, block, lds, stream, K, (double *)nullptr, 0, Q); \
            break
        switch (p->d) {
            TPR_PCRS_CASE(1); TPR_PCRS_CASE(2); TPR_PCRS_CASE(3); TPR_PCRS_CASE(4);
            TPR_PCRS_CASE(5); TPR_PCRS_CASE(6); TPR_PCRS_CASE(7); TPR_PCRS_CASE(8);
            TPR_PCRS_CASE(9); TPR_PCRS_CASE(10); TPR_PCRS_CASE(11); TPR_PCRS_CASE(12);
            TPR_PCRS_CASE(13); TPR_PCRS_CASE(14); TPR_PCRS_CASE(15); TPR_PCRS_CASE(16);
        }
#undef TPR_PCRS_CASE
    }
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_ppoly_eval_batch(int B, int nseg, int d, const double *coef, const double *breaks, const int32_t *counts, int T,
                         const double *times, int order, double *out, int device_ptrs, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (B < 0 || nseg < 1 || d < 1 || T < 0 || order < 0 || order > 2 || !coef || !breaks || !times || !out)
        return fail(TPR_E_BADARG, "bad piecewise-polynomial evaluation arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(device_ptrs != 0, coef));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(device_ptrs != 0, stream);
    tpr::PpolyArgs A{};
    A.B = B; A.nseg = nseg; A.d = d; A.T = T; A.order = order;
    A.coef = S.in(coef, (size_t)B * 4 * nseg * d);
    A.breaks = S.in(breaks, (size_t)B * (nseg + 1));
    A.counts = S.in(counts, (size_t)B);
    A.times = S.in(times, (size_t)B * T);
    A.out = S.out(out, (size_t)B * T * d);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    const long long total = (long long)B * T * d;  // one thread per (sample, dof)
    if (total > (long long)0x7fffffff * 256) return fail(TPR_E_BADARG, "piecewise-polynomial evaluation: B T d too large for one launch");
    // (round 3 also tried a block per path with the breakpoints searched in LDS on the thread-per-sample form: 1.07 ->
    // 1.20 ms -- the scattered coefficient rows were what it waited for, not the search)
    if (total > 0)
        hipLaunchKernelGGL(tpr::ppoly_eval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_lp1d_batch(int n, int nrows, const double *v, const double *a, const double *b,
                   const double *low, const double *high, int32_t *result, double *optval,
                   double *optvar, int32_t *active, void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (n < 0 || nrows < 0 || !v || !low || !high || !result || !optval || !optvar || !active)
        return fail(TPR_E_BADARG, "bad lp1d arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(false, nullptr));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(false, stream);
    const size_t nn = (size_t)n, rows = nn * (size_t)nrows;
    const double *dv = S.in(v, nn * 2), *da = S.in(a, rows), *db = S.in(b, rows);
    const double *dlow = S.in(low, nn), *dhigh = S.in(high, nn);
    int32_t *dres = S.out(result, nn), *dact = S.out(active, nn);
    double *dval = S.out(optval, nn), *dvar = S.out(optvar, nn);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (n > 0)
        hipLaunchKernelGGL(tpr::lp1d_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, nrows, dv, da, db,
                           dlow, dhigh, dres, dval, dvar, dact);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

int tpr_lp2d_batch(int n, int nrows, const double *v, const double *a, const double *b,
                   const double *c, const double *low, const double *high, const int32_t *active_in,
                   int32_t *result, double *optval, double *optvar, int32_t *active_out,
                   void *stream_) {
    if (g_device < 0) return fail(TPR_E_HIP, "tpr_init() has not succeeded");
    if (n < 0 || nrows < 0 || nrows > tpr::kKatMaxRows || !v || !low || !high || !active_in || !result ||
        !optval || !optvar || !active_out)
        return fail(TPR_E_BADARG, "bad lp2d arguments (nrows <= 128)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceScope scope(call_device(false, nullptr));
    if (scope.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(scope.err));
    Staging S(false, stream);
    const size_t nn = (size_t)n, rows = nn * (size_t)nrows;
    const double *dv = S.in(v, nn * 3), *da = S.in(a, rows), *db = S.in(b, rows), *dc = S.in(c, rows);
    const double *dlow = S.in(low, nn * 2), *dhigh = S.in(high, nn * 2);
    const int32_t *dain = S.in(active_in, nn * 2);
    int32_t *dres = S.out(result, nn), *daout = S.out(active_out, nn * 2);
    double *dval = S.out(optval, nn), *dvar = S.out(optvar, nn * 2);
    if (S.err != hipSuccess) return fail(TPR_E_HIP, hipGetErrorString(S.err));
    if (n > 0)
        hipLaunchKernelGGL(tpr::lp2d_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, n, nrows, dv, da, db,
                           dc, dlow, dhigh, dain, dres, dval, dvar, daout);
    HIP_TRY(S.finish());
    return TPR_E_OK;
}

}  // extern "C"

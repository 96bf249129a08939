// tpr_cert_tu.hip -- translation unit of kernel family 3 (tpr_cert.hip.inc) for ONE dof.
//
// The certified lane kernels are instantiated per dof and per (sd output, grid in LDS, discretisation, certificate mode):
// 8 x 16 heavy kernels, most of the library's compile time.  build.py compiles this file once per dof
// (-DTPR_TU_D=<dof>), in parallel with tpr_kernels.hip, and links the objects into libtoppra_hip.so; the entry points
// below are the only interface (declared in tpr_kernels.hip).  Development builds with instrumentation defines
// (-DTPR_DEBUG_PREDICT ..., whose counters are device globals of ONE translation unit) include this file from
// tpr_kernels.hip instead (TPR_SINGLE_TU).
#include <hip/hip_runtime.h>

#include "../../include/toppra_hip.h"
#include "tpr_device.hpp"
#include "tpr_group.hip.inc"
#include "tpr_cert.hip.inc"

#ifndef TPR_TU_D
#error "compile with -DTPR_TU_D=<dof 1..14>"
#endif
// A dof's unit can be split by entry point (build.py::CERT_UNIT_PARTS): 1 = the fused solve / backward scan, 2 = feasible sets,
// 3 = TOPPRAsd; 0 = all three in one unit.  Above 8 dof the three kernels want DIFFERENT scheduler flags (the switch that takes
// the 13-dof solve from 8.0 to 5.2 ms takes its feasible-sets kernel from 9.3 to 10.8), and flags are per translation unit.
#ifndef TPR_TU_PART
#define TPR_TU_PART 0
#endif

// The certificates follow the reference's whole pivot trace (tpr_cert_lane.hip.inc: cert_propose_sound & co) -- the only mode
// of the product at every dof this file is compiled for (1..13; slim blocks above 8 dof); TPR_SOUND_CERTIFICATES is accepted
// and changes nothing.  The round-2/3 "fast" certificates (last pivot only) survive in the opt-in tolerance measurement
// build.  History of the 9..13-dof instantiations (round 3: not shipped sound; round 4: one conditionally-needed load in
// CertStage::fetch): DESIGN.md sections 3.2 and 9; tests/test_kernel_resources.py pins their scratch and branch counts.
#ifdef TPR_TOLERANCE_MODE
constexpr bool kSoundKernels = false;
#else
constexpr bool kSoundKernels = true;
#endif

// The grid-in-LDS instantiations exist up to 8 dof only (above, the grid is read from global memory: see the launchers): a
// kernel that is never launched is not compiled either -- half the compile time of the 9..13-dof units, and nothing unlaunched
// for the build's code-generation check (toppra_amd/codegen_check.py) to trip over.
#define TPR_IF_GRID_LDS(T, F)                                   \
    do {                                                        \
        if constexpr (D <= 8) { if (grid_lds) { T; } else { F; } } \
        else { F; }                                             \
    } while (0)

// Slim blocks (above 8 dof): the per-block transposed workspace of GroupArgs::tws, stream-ordered around the launch.
struct TwsScope {
    void *ws = nullptr;
    hipStream_t stream;
    hipError_t err = hipSuccess;
    TwsScope(tpr::GroupArgs &G, int d, int blocks, hipStream_t st) : stream(st) {
        if (d <= 8) return;
        err = hipMallocAsync(&ws, (size_t)blocks * tpr::cert_tws_fields(d, G.nseg) * 64 * sizeof(double), st);
        G.tws = static_cast<double *>(ws);
    }
    ~TwsScope() { if (ws) (void)hipFreeAsync(ws, stream); }
};

#define TPR_TU_CAT2(a, b) a##b
#define TPR_TU_CAT(a, b) TPR_TU_CAT2(a, b)

#if TPR_TU_PART == 0 || TPR_TU_PART == 1
// Launch the certified lane kernel for TPR_TU_D dof: the fused solve, or the backward scan alone (G.backward_only).
// Returns 0, or -1 when the launch geometry cannot be met (never for supported shapes).
extern "C" __attribute__((visibility("hidden"))) int TPR_TU_CAT(tpr_tu_cert_launch_, TPR_TU_D)(const tpr::GroupArgs *Gp, hipStream_t stream) {
    constexpr int D = TPR_TU_D, BS = 64;
    tpr::GroupArgs G = *Gp;
    const dim3 grid((G.B + BS - 1) / BS), block(BS);
    TwsScope tws(G, D, (int)grid.x, stream);
    if (tws.err != hipSuccess) return -1;
    // the shared grid goes to LDS only while four blocks still fit a CU (160 KB): a fifth of the
    // 1024 blocks of a 65536-trajectory batch would otherwise wait for a second round
    const size_t grid_bytes = (size_t)(G.N + 1) * sizeof(double);
    using CS = tpr::CertStage<D, BS>;
    const size_t static_lds = (((4 * D + CS::kLimCols) > 24 ? (4 * D + CS::kLimCols) : 24) * BS + tpr::kCertXch * BS +
                               tpr::cert_batch_groups<D>() * tpr::GroupCfg<D, tpr::kCertBatchLanes>::kRowBuf + CS::kRing * 2 * BS) * sizeof(double);
    // the shared grid goes to LDS only while that does not cost a block per CU (160 KB: four blocks up to 8 dof -- one
    // wave per SIMD --, three at 9..11 dof, two above)
    // ... and only up to 8 dof (round 3: the <9 dof, grid in LDS, sound certificates> instantiation returned lower bounds that
    // were off in the last bits on 0.9 % of an irregular batch while value-identical spellings were bit-exact -- the signature of
    // the register allocator's copy above the exec restore, profiles/r06_miscompile_root_cause.md, which the build now checks
    // for); above 8 dof the grid is read from global memory, and those instantiations are not compiled (TPR_IF_GRID_LDS).
    const bool grid_lds = D <= 8 && !(G.flags & TPR_GRID_PER_TRAJ) && (160 * 1024) / (static_lds + grid_bytes) == (160 * 1024) / static_lds;
    const size_t lds = grid_lds ? grid_bytes : 0;
    // One 64-lane block per wave; ~39 KB of LDS per block leaves one wave per SIMD, which the kernel
    // is written for (the whole register file, stalls covered by unrolled independent row work).
#define TPR_LAUNCH_CERT(SD, GL, IN, SO) hipLaunchKernelGGL((tpr::cert_solve_kernel<D, BS, SD, GL, IN, SO>), grid, block, lds, stream, G)
#define TPR_LAUNCH_CERT3(SD, GL, IN) TPR_LAUNCH_CERT(SD, GL, IN, kSoundKernels)
    if (G.flags & TPR_ACC_INTERPOLATION) {
        if (G.sd) TPR_IF_GRID_LDS(TPR_LAUNCH_CERT3(true, true, true), TPR_LAUNCH_CERT3(true, false, true));
        else TPR_IF_GRID_LDS(TPR_LAUNCH_CERT3(false, true, true), TPR_LAUNCH_CERT3(false, false, true));
    } else {  // Collocation
        if (G.sd) TPR_IF_GRID_LDS(TPR_LAUNCH_CERT3(true, true, false), TPR_LAUNCH_CERT3(true, false, false));
        else TPR_IF_GRID_LDS(TPR_LAUNCH_CERT3(false, true, false), TPR_LAUNCH_CERT3(false, false, false));
    }
#undef TPR_LAUNCH_CERT3
#undef TPR_LAUNCH_CERT
    return 0;
}
#endif

#if TPR_TU_PART == 0 || TPR_TU_PART == 2
// compute_feasible_sets on the certified lane design (cert_feasible_kernel): X [B][N+1][2].
extern "C" __attribute__((visibility("hidden"))) int TPR_TU_CAT(tpr_tu_cert_feasible_launch_, TPR_TU_D)(const tpr::GroupArgs *Gp, double *X, hipStream_t stream) {
    constexpr int D = TPR_TU_D, BS = 64;
    tpr::GroupArgs G = *Gp;
    const dim3 grid((G.B + BS - 1) / BS), block(BS);
    TwsScope tws(G, D, (int)grid.x, stream);
    if (tws.err != hipSuccess) return -1;
    const size_t grid_bytes = (size_t)(G.N + 1) * sizeof(double);
    using CS = tpr::CertStage<D, BS>;
    const size_t static_lds = (((4 * D + CS::kLimCols) > 24 ? (4 * D + CS::kLimCols) : 24) * BS + tpr::kCertXch * BS +
                               tpr::cert_batch_groups<D>() * tpr::GroupCfg<D, tpr::kCertBatchLanes>::kRowBuf + CS::kRing * 2 * BS) * sizeof(double);
    // the shared grid goes to LDS only while that does not cost a block per CU (160 KB: four blocks up to 8 dof -- one
    // wave per SIMD --, three at 9..11 dof, two above)
    // ... and only up to 8 dof (round 3: the <9 dof, grid in LDS, sound certificates> instantiation returned lower bounds that
    // were off in the last bits on 0.9 % of an irregular batch while value-identical spellings were bit-exact -- the signature of
    // the register allocator's copy above the exec restore, profiles/r06_miscompile_root_cause.md, which the build now checks
    // for); above 8 dof the grid is read from global memory, and those instantiations are not compiled (TPR_IF_GRID_LDS).
    const bool grid_lds = D <= 8 && !(G.flags & TPR_GRID_PER_TRAJ) && (160 * 1024) / (static_lds + grid_bytes) == (160 * 1024) / static_lds;
    const size_t lds = grid_lds ? grid_bytes : 0;
    const bool interp = (G.flags & TPR_ACC_INTERPOLATION) != 0;
#define TPR_LAUNCH_FEAS(GL, IN, SO) hipLaunchKernelGGL((tpr::cert_feasible_kernel<D, BS, GL, IN, SO>), grid, block, lds, stream, G, X)
#define TPR_LAUNCH_FEAS2(GL, IN) TPR_LAUNCH_FEAS(GL, IN, kSoundKernels)
    if (interp) TPR_IF_GRID_LDS(TPR_LAUNCH_FEAS2(true, true), TPR_LAUNCH_FEAS2(false, true));
    else TPR_IF_GRID_LDS(TPR_LAUNCH_FEAS2(true, false), TPR_LAUNCH_FEAS2(false, false));
#undef TPR_LAUNCH_FEAS2
#undef TPR_LAUNCH_FEAS
    return 0;
}
#endif

#if TPR_TU_PART == 0 || TPR_TU_PART == 3
// TOPPRAsd: backward scan + the fastest / slowest forward profiles in one launch (cert_solve_kernel<..., SDFWD = true>).
extern "C" __attribute__((visibility("hidden"))) int TPR_TU_CAT(tpr_tu_cert_sd_launch_, TPR_TU_D)(const tpr::GroupArgs *Gp, hipStream_t stream) {
    constexpr int D = TPR_TU_D, BS = 64;
    tpr::GroupArgs G = *Gp;
    const dim3 grid((G.B + BS - 1) / BS), block(BS);
    TwsScope tws(G, D, (int)grid.x, stream);
    if (tws.err != hipSuccess) return -1;
    const size_t grid_bytes = (size_t)(G.N + 1) * sizeof(double);
    using CS = tpr::CertStage<D, BS>;
    const size_t cols = ((4 * D + CS::kLimCols) > 32 ? (4 * D + CS::kLimCols) : 32) * BS;
    const size_t static_lds = (cols + tpr::kCertXch * BS + tpr::cert_batch_groups<D>() * tpr::GroupCfg<D, tpr::kCertBatchLanes>::kRowBuf + CS::kRing * 2 * BS) * sizeof(double);
    // the shared grid goes to LDS only while that does not cost a block per CU (160 KB: four blocks up to 8 dof -- one
    // wave per SIMD --, three at 9..11 dof, two above)
    // ... and only up to 8 dof (round 3: the <9 dof, grid in LDS, sound certificates> instantiation returned lower bounds that
    // were off in the last bits on 0.9 % of an irregular batch while value-identical spellings were bit-exact -- the signature of
    // the register allocator's copy above the exec restore, profiles/r06_miscompile_root_cause.md, which the build now checks
    // for); above 8 dof the grid is read from global memory, and those instantiations are not compiled (TPR_IF_GRID_LDS).
    const bool grid_lds = D <= 8 && !(G.flags & TPR_GRID_PER_TRAJ) && (160 * 1024) / (static_lds + grid_bytes) == (160 * 1024) / static_lds;
    const size_t lds = grid_lds ? grid_bytes : 0;
    const bool interp = (G.flags & TPR_ACC_INTERPOLATION) != 0;
#define TPR_LAUNCH_SD(GL, IN, SO) hipLaunchKernelGGL((tpr::cert_solve_kernel<D, BS, false, GL, IN, SO, true>), grid, block, lds, stream, G)
#define TPR_LAUNCH_SD2(GL, IN) TPR_LAUNCH_SD(GL, IN, kSoundKernels)
    if (interp) TPR_IF_GRID_LDS(TPR_LAUNCH_SD2(true, true), TPR_LAUNCH_SD2(false, true));
    else TPR_IF_GRID_LDS(TPR_LAUNCH_SD2(true, false), TPR_LAUNCH_SD2(false, false));
#undef TPR_LAUNCH_SD2
#undef TPR_LAUNCH_SD
    return 0;
}
#endif


// tpr_robust_tu.hip -- translation unit of the robust (conic) kernels (tpr_robust.hip.inc).
//
// build.py compiles this file twice (-DTPR_TU_HALF=0: the generic lane kernel and 1..8 dof on 8 lanes per trajectory;
// =1: 9..16 dof on 16 lanes), in parallel with the other units; instrumented single-unit builds include it from
// tpr_kernels.hip with TPR_TU_HALF = 2 (both halves).  One entry point per half, declared in tpr_kernels.hip.
#include <hip/hip_runtime.h>

#include "../../include/toppra_hip.h"
#include "tpr_device.hpp"
#include "tpr_lane.hip.inc"
#include "tpr_group.hip.inc"
#include "tpr_robust.hip.inc"

#ifndef TPR_TU_HALF
#error "compile with -DTPR_TU_HALF=0|1"
#endif

namespace {
template <int D, int L>
int robust_launch_group(const tpr::RobustArgs &P, size_t max_lds, hipStream_t stream) {
    using C = tpr::GroupCfg<D, L>;
    const tpr::BatchArgs &A = P.A;
    auto lds_bytes = [&](int threads) { return (size_t)(threads / L) * C::lds_doubles(A.nseg, true) * sizeof(double); };
    if (lds_bytes(64) > max_lds) return 1;  // very long spline tables: the caller falls back to the lane kernel
    int threads = 64;
    for (int t = 256; t > 64; t /= 2)
        if (lds_bytes(t) <= max_lds) { threads = t; break; }
    while (threads > 64 && (long long)A.B * L / threads < 4 * 256) threads /= 2;
    const int groups = threads / L;
    hipLaunchKernelGGL((tpr::group_robust_kernel<D, L>), dim3((A.B + groups - 1) / groups), dim3(threads), lds_bytes(threads), stream, P);
    return 0;
}
}  // namespace

#if TPR_TU_HALF == 0 || TPR_TU_HALF == 2
// 0 = launched; 1 = this shape needs the lane kernel (tpr_tu_robust_lane_launch); -1 = dof not served here
extern "C" __attribute__((visibility("hidden"))) int tpr_tu_robust_launch_lo(const tpr::RobustArgs *P, size_t max_lds, hipStream_t stream) {
    switch (P->A.d) {
        case 1: return robust_launch_group<1, 8>(*P, max_lds, stream);
        case 2: return robust_launch_group<2, 8>(*P, max_lds, stream);
        case 3: return robust_launch_group<3, 8>(*P, max_lds, stream);
        case 4: return robust_launch_group<4, 8>(*P, max_lds, stream);
        case 5: return robust_launch_group<5, 8>(*P, max_lds, stream);
        case 6: return robust_launch_group<6, 8>(*P, max_lds, stream);
        case 7: return robust_launch_group<7, 8>(*P, max_lds, stream);
        case 8: return robust_launch_group<8, 8>(*P, max_lds, stream);
    }
    return -1;
}
extern "C" __attribute__((visibility("hidden"))) int tpr_tu_robust_lane_launch(const tpr::RobustArgs *P, hipStream_t stream) {
    hipLaunchKernelGGL(tpr::robust_solve_kernel, dim3((P->A.B + 63) / 64), dim3(64), 0, stream, *P);
    return 0;
}
#endif
#if TPR_TU_HALF == 1 || TPR_TU_HALF == 2
extern "C" __attribute__((visibility("hidden"))) int tpr_tu_robust_launch_hi(const tpr::RobustArgs *P, size_t max_lds, hipStream_t stream) {
    switch (P->A.d) {
        case 9: return robust_launch_group<9, 16>(*P, max_lds, stream);
        case 10: return robust_launch_group<10, 16>(*P, max_lds, stream);
        case 11: return robust_launch_group<11, 16>(*P, max_lds, stream);
        case 12: return robust_launch_group<12, 16>(*P, max_lds, stream);
        case 13: return robust_launch_group<13, 16>(*P, max_lds, stream);
        case 14: return robust_launch_group<14, 16>(*P, max_lds, stream);
        case 15: return robust_launch_group<15, 16>(*P, max_lds, stream);
        case 16: return robust_launch_group<16, 16>(*P, max_lds, stream);
    }
    return -1;
}
#endif

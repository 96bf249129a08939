// tpr_robust_args.hpp -- argument block of the robust (conic) kernels (tpr_robust.hip.inc), shared with the dispatcher.
#pragma once
namespace tpr {
struct RobustArgs {
    BatchArgs A;
    double ru, rx, rc;
    double *X;  // feasible sets [B][N+1][2] or nullptr
};
}  // namespace tpr

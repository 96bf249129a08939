// tpr_dense_args.hpp -- argument block of the dense-row kernels (tpr_dense.hip.inc), shared with the dispatcher.
#pragma once
#include <cstdint>
namespace tpr {
struct DenseArgs {
    int B, N, nC, flags;
    const double *a, *b, *c;     // [B][N+1][nC]
    const double *low, *high;    // [B][N+1][2]  (u, x) boxes
    const double *deltas;        // [B][N]
    const double *sd_start, *sd_end, *sd_end_hi;   // [B] (sd_end_hi: controllable sets from an interval)
    double *sd2, *sd, *u, *K;    // as tpr_result
    int32_t *status;
    double *X;                   // feasible sets [B][N+1][2] (mode 2)
    int backward_only;
    int32_t *active;             // [B][4] warm-start state of the wrapper object (active_c_up[2], active_c_down[2]), in / out; nullptr: a fresh object
    // TOPPRAsd: the fastest / slowest forward profiles x [B][N+1], u [B][N] (dense_sd_forward_kernel reads K and status)
    double *sd_xf, *sd_uf, *sd_xl, *sd_ul;
};
}  // namespace tpr

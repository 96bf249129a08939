// tpr_dense_tu.hip -- translation unit of the dense-row kernels (tpr_dense.hip.inc): the reference's seidelWrapper contract
// for ANY canonical-linear constraint set.  build.py compiles it in parallel with the other units; instrumented
// single-unit builds include it from tpr_kernels.hip.  One entry point, declared in tpr_kernels.hip.
#include <hip/hip_runtime.h>

#include "../../include/toppra_hip.h"
#include "tpr_device.hpp"
#include "tpr_group.hip.inc"
#include "tpr_dense_args.hpp"
#include "tpr_dense.hip.inc"

namespace {
template <int D, int L>
int dense_launch(const tpr::DenseArgs &A, int feasible /* 0 solve, 1 feasible sets, 2 TOPPRAsd forward scans */, hipStream_t stream) {
    using C = tpr::GroupCfg<D, L>;
    int threads = 256;
    while (threads > 64 && (long long)A.B * L / threads < 4 * 256) threads /= 2;  // small batches: more, smaller blocks
    const int groups = threads / L;
    const size_t lds = (size_t)groups * C::kRowBuf * sizeof(double);
    const dim3 grid((A.B + groups - 1) / groups), block(threads);
    if (feasible == 1) hipLaunchKernelGGL((tpr::dense_feasible_kernel<D, L>), grid, block, lds, stream, A);
    else if (feasible == 2) hipLaunchKernelGGL((tpr::dense_sd_forward_kernel<D, L>), grid, block, lds, stream, A);
    else hipLaunchKernelGGL((tpr::dense_solve_kernel<D, L>), grid, block, lds, stream, A);
    return 0;
}
}  // namespace

// nC rows per stage (incl. the two x_next rows) -> the smallest row-slot layout that holds them: 2 + 4 D >= nC.
// 0 = launched, -1 = more rows than the layouts hold (122: the row keys carry the virtual row index in 7 bits).
extern "C" __attribute__((visibility("hidden"))) int tpr_tu_dense_launch(const tpr::DenseArgs *A, int feasible, hipStream_t stream) {
    const int D = A->nC <= 6 ? 1 : (A->nC - 2 + 3) / 4;
    switch (D) {
#define TPR_DENSE_CASE(DD, LL) case DD: return dense_launch<DD, LL>(*A, feasible, stream)
        TPR_DENSE_CASE(1, 8); TPR_DENSE_CASE(2, 8); TPR_DENSE_CASE(3, 8); TPR_DENSE_CASE(4, 8);
        TPR_DENSE_CASE(5, 8); TPR_DENSE_CASE(6, 8); TPR_DENSE_CASE(7, 8); TPR_DENSE_CASE(8, 8);
        TPR_DENSE_CASE(9, 16); TPR_DENSE_CASE(10, 16); TPR_DENSE_CASE(11, 16); TPR_DENSE_CASE(12, 16);
        TPR_DENSE_CASE(13, 16); TPR_DENSE_CASE(14, 16); TPR_DENSE_CASE(15, 16); TPR_DENSE_CASE(16, 16);
        TPR_DENSE_CASE(17, 32); TPR_DENSE_CASE(18, 32); TPR_DENSE_CASE(19, 32); TPR_DENSE_CASE(20, 32);
        TPR_DENSE_CASE(21, 32); TPR_DENSE_CASE(22, 32); TPR_DENSE_CASE(23, 32); TPR_DENSE_CASE(24, 32);
        TPR_DENSE_CASE(25, 32); TPR_DENSE_CASE(26, 32); TPR_DENSE_CASE(27, 32); TPR_DENSE_CASE(28, 32);
        TPR_DENSE_CASE(29, 32); TPR_DENSE_CASE(30, 32);
#undef TPR_DENSE_CASE
    }
    return -1;
}

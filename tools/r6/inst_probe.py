"""Round-6 forensics: every instantiation of kernel family 3 at one dof against the rows-across-lanes kernels; ONE summary line
(which of solve / feasible sets / TOPPRAsd x grid x discretisation x sd output differ, and in which outputs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toppra_amd import batch, _capi
from tests.test_gpu_instantiations import _problem
_capi.init(0)
d = int(sys.argv[1])
bad = []
def cmp(tag, a, b, keys):
    diff = [k for k in keys if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
    if diff:
        bad.append(tag + ":" + "+".join(diff))
for ptg in (False, True):
    data, grid, sd0, sd1 = _problem(d, 700 + d, ptg)
    desired = np.random.default_rng(d).uniform(0.5, 5.0, size=grid.shape[0] if grid.ndim == 2 else 96)
    for interp in (True, False):
        for want_sd in (False, True):
            args = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"], sd0, sd1, interp)
            cmp("solve[g%d i%d s%d]" % (ptg, interp, want_sd), batch.solve_batch(*args, want_sd=want_sd, variant=2),
                batch.solve_batch(*args, want_sd=want_sd, variant=3), ("K", "sd2", "u", "status"))
        X2 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=2)
        X3 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=3)
        cmp("feas[g%d i%d]" % (ptg, interp), {"X": X2}, {"X": X3}, ("X",))
        a = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1, variant=2, interpolation=interp)
        b = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1, variant=3, interpolation=interp)
        cmp("sd[g%d i%d]" % (ptg, interp), a, b, ("K", "sd2", "u", "alpha", "status"))
print("PASS" if not bad else "FAIL " + " ".join(bad))

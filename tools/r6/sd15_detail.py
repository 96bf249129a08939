"""Round-6: where exactly does TOPPRAsd of family 3 leave family 2 at 15 dof (tests/test_gpu_instantiations' problem)?"""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from toppra_amd import batch, _capi
from tests.test_gpu_instantiations import _problem
_capi.init(0)
np.set_printoptions(precision=17, linewidth=200)
d = 15
data, grid, sd0, sd1 = _problem(d, 700 + d, False)
desired = np.random.default_rng(d).uniform(0.5, 5.0, size=96)
args = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1)
a = batch.solve_desired_duration_batch(*args, variant=2, interpolation=True)
b = batch.solve_desired_duration_batch(*args, variant=3, interpolation=True)
bad = [t for t in range(96) if not (np.array_equal(a["sd2"][t], b["sd2"][t], equal_nan=True) and np.array_equal(a["u"][t], b["u"][t], equal_nan=True))]
print("trajectories off:", bad, "alpha of those:", [float(a["alpha"][t]) for t in bad][:8])
for t in bad[:3]:
    print("trajectory", t)
    for i in range(grid.shape[-1] - 1):
        xr, xb, ur, ub = a["sd2"][t, i], b["sd2"][t, i], a["u"][t, i], b["u"][t, i]
        if xr != xb or ur != ub:
            print("  stage %2d  x ref %.17g got %.17g (rel %.2e)   u ref %.17g got %.17g (rel %.2e)" % (i, xr, xb, (xb - xr) / xr if xr else 0, ur, ub, (ub - ur) / abs(ur) if ur else 0))

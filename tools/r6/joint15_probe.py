import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from toppra_amd import batch, _capi
_capi.init(0)
from tests.test_gpu_instantiations import _tight_joint_problem
d = 15
data = _tight_joint_problem(d, 900 + d)
desired = np.random.default_rng(50 + d).uniform(2.0, 40.0, size=4 * d)
a = batch.solve_desired_duration_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired, None, None, variant=2, interpolation=True)
b = batch.solve_desired_duration_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired, None, None, variant=3, interpolation=True)
for k in ("status", "K", "sd2", "u", "alpha"):
    x, y = np.asarray(a[k], float), np.asarray(b[k], float)
    ne = ~((x == y) | (np.isnan(x) & np.isnan(y)))
    print(k, int(ne.sum()), "entries on trajectories", sorted(set(np.argwhere(ne)[:, 0].tolist())))
print("(trajectory j: joint j % 15, kind j // 15: 0 both limits tight, 1 amax only, 2 amin only, 3 both + velocity)")

"""One line: ONE entry point of family 3 at one dof on the joint-binding batch (tests/test_gpu_instantiations._tight_joint_problem) and on the
instantiation test's random batch, against family 2 -- PASS / FAIL."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from toppra_amd import batch, _capi
_capi.init(0)
from tests.test_gpu_instantiations import _tight_joint_problem, _problem
d, part = int(sys.argv[1]), int(sys.argv[2])
bad = []
def eq(a, b, keys, tag):
    for k in keys:
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True): bad.append(tag + ":" + k)
for interp in (True, False):
    for name, (data, grid, sd0, sd1) in (("joint", (lambda D: (D, D["grid"], None, None))(_tight_joint_problem(d, 900 + d))), ("random", _problem(d, 800 + d, False)), ("random-g", _problem(d, 800 + d, True))):
        Bn = data["coef"].shape[0]
        base = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"])
        tag = "%s/i%d" % (name, interp)
        if part == 1:
            for want_sd in (False, True):
                eq(batch.solve_batch(*base, sd0, sd1, interp, want_sd=want_sd, variant=3), batch.solve_batch(*base, sd0, sd1, interp, want_sd=want_sd, variant=2), ("status", "K", "sd2", "u"), tag)
        elif part == 2:
            eq({"X": batch.feasible_sets_batch(*base, interp, variant=3)}, {"X": batch.feasible_sets_batch(*base, interp, variant=2)}, ("X",), tag)
        else:
            desired = np.random.default_rng(50 + d).uniform(0.5, 40.0, size=Bn)
            eq(batch.solve_desired_duration_batch(*base, desired, sd0, sd1, variant=3, interpolation=interp),
               batch.solve_desired_duration_batch(*base, desired, sd0, sd1, variant=2, interpolation=interp), ("status", "K", "sd2", "sd", "u", "alpha"), tag)
print("joint+inst PASS" if not bad else "joint+inst FAIL " + " ".join(bad[:6]))

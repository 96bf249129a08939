"""One line: the time of ONE entry point of family 3 at one dof (part 1 = solve kernel, 2 = feasible sets per call, 3 = TOPPRAsd per
call; 65536 x d x 200) + parity of that entry with family 2 on 2048 trajectories."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
_capi.init(0)
d, part = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
data = tb.make_synthetic_batch(65536, d, 200)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
small = tb.make_synthetic_batch(2048, d, 200, seed=5)
args = [small[k] for k in ("coef", "breaks", "grid", "vlim", "alim")]
def wall(fn):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(8): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
    return best
eq = lambda a, b, keys: all(np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True) for k in keys)
if part == 1:
    out = tb.solve_batch(*dv, variant=3); torch.cuda.synchronize()
    ms = min(tb.solve_batch_timed(*dv, out, reps=5, variant=3) for _ in range(3))
    ok = eq(tb.solve_batch(*args, variant=3), tb.solve_batch(*args, variant=2), ("K", "sd2", "u", "status"))
elif part == 2:
    ms = wall(lambda: tb.feasible_sets_batch(*dv, variant=3))
    ok = np.array_equal(tb.feasible_sets_batch(*args, variant=3), tb.feasible_sets_batch(*args, variant=2), equal_nan=True)
else:
    ms = wall(lambda: tb.solve_desired_duration_batch(*dv, 3.0, variant=3))
    des = np.random.default_rng(3).uniform(0.5, 6.0, 2048)
    ok = eq(tb.solve_desired_duration_batch(*args, des, variant=3), tb.solve_desired_duration_batch(*args, des, variant=2), ("K", "sd2", "sd", "u", "alpha", "status"))
print("part %d: %.3f ms  parity %s" % (part, ms, "PASS" if ok else "FAIL"))

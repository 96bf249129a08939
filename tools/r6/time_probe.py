"""One line: kernel times of family 3 at one dof (solve, feasible sets, TOPPRAsd; 65536 x d x 200) + parity against family 2."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
d = int(sys.argv[1])
dev = torch.device("cuda", 0)
data = tb.make_synthetic_batch(65536, d, 200)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
out = tb.solve_batch(*dv, variant=3); torch.cuda.synchronize()
ms = min(tb.solve_batch_timed(*dv, out, reps=5, variant=3) for _ in range(3))
def wall(fn):
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); [fn() for _ in range(3)]; ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / 3
fs = wall(lambda: tb.feasible_sets_batch(*dv, variant=3))
sd = wall(lambda: tb.solve_desired_duration_batch(*dv, 3.0, variant=3))
small = tb.make_synthetic_batch(2048, d, 200, seed=5)
args = [small[k] for k in ("coef", "breaks", "grid", "vlim", "alim")]
a, b = tb.solve_batch(*args, variant=3), tb.solve_batch(*args, variant=2)
ok = all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("K", "sd2", "u", "status"))
print("solve %.3f ms  feasible %.3f ms  TOPPRAsd %.3f ms  parity %s" % (ms, fs, sd, "PASS" if ok else "FAIL"))

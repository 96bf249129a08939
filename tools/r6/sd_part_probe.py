"""One line: TOPPRAsd on family 3 at one dof (65536 x d x 200, desired 3 s; best of three 10-call averages) + parity with family 2."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
_capi.init(0)
d = int(sys.argv[1])
dev = torch.device("cuda", 0)
data = tb.make_synthetic_batch(65536, d, 200)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
fn = lambda: tb.solve_desired_duration_batch(*dv, 3.0, variant=3)
fn(); fn(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
small = tb.make_synthetic_batch(2048, d, 200, seed=5)
args = [small[k] for k in ("coef", "breaks", "grid", "vlim", "alim")]
des = np.random.default_rng(3).uniform(0.5, 6.0, 2048)
a, b = tb.solve_desired_duration_batch(*args, des, variant=3), tb.solve_desired_duration_batch(*args, des, variant=2)
ok = all(np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True) for k in ("K", "sd2", "sd", "u", "alpha", "status"))
print("TOPPRAsd %.3f ms  parity %s" % (best, "PASS" if ok else "FAIL"))

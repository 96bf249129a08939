"""One line: kernel times of the families that live in tpr_kernels.hip (5, 4, 2, robust) at their own shapes + parity of each with family 3 /
the oracle on a small batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
_capi.init(0)
dev = torch.device("cuda", 0)
def dvof(B, d, N, seed=None):
    data = tb.make_synthetic_batch(B, d, N) if seed is None else tb.make_synthetic_batch(B, d, N, seed=seed)
    return [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
def t(B, d, N, v, reps=5):
    dv = dvof(B, d, N)
    out = tb.solve_batch(*dv, variant=v); torch.cuda.synchronize()
    return min(tb.solve_batch_timed(*dv, out, reps=reps, variant=v) for _ in range(3))
res = ["c2_v5 %.3f" % t(4096, 7, 200, 5), "c2_v4 %.3f" % t(4096, 7, 200, 4), "c1_v4 %.4f" % t(1, 7, 100, 4, reps=20),
       "f2_32768x7 %.3f" % t(32768, 7, 200, 2), "f2_65536x7 %.3f" % t(65536, 7, 200, 2), "f2_65536x14 %.3f" % t(65536, 14, 200, 2, reps=2),
       "f2_65536x16 %.3f" % t(65536, 16, 200, 2, reps=2)]
dv4 = dvof(16384, 7, 100)
def wall(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
res.append("robust_c4 %.3f" % wall(lambda: tb.robust_solve_batch(*dv4, [1e-3, 5e-2, 9e-3])))
small = tb.make_synthetic_batch(2048, 7, 200, seed=5)
args = [small[k] for k in ("coef", "breaks", "grid", "vlim", "alim")]
ref = tb.solve_batch(*args, variant=3)
ok = all(all(np.array_equal(tb.solve_batch(*args, variant=v)[k], ref[k], equal_nan=True) for k in ("K", "sd2", "u", "status")) for v in (2, 4, 5))
print("  ".join(res), " parity", "PASS" if ok else "FAIL")

"""TOPPRAsd at the headline shape (65536 x 7 x 200, device tensors in and out; another dof as argv[1]): wall time per call, the share of trajectories that
bisect, and parity of the fused path (family 3: durations summed in the forward scans) with the rows-across-lanes kernels
(sd_finish_kernel's own duration passes) on a sub-batch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb, _capi
_capi.init(0)
B, d, N = 65536, (int(sys.argv[1]) if len(sys.argv) > 1 else 7), 200
data = tb.make_synthetic_batch(B, d, N)
dev = torch.device("cuda", 0)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
for desired in (3.0, None):
    des = 3.0 if desired is not None else torch.from_numpy(np.random.default_rng(5).uniform(0.5, 6.0, B)).to(dev)
    fn = lambda: tb.solve_desired_duration_batch(*dv, des)
    out = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    al = out["alpha"].cpu().numpy()
    print("desired %s: %.3f ms per call; alpha = 1: %.1f %%, alpha = 0: %.1f %%, bisected: %.1f %%" % (
        "3.0" if desired is not None else "U(0.5, 6)", ms, 100 * np.mean(al == 1.0), 100 * np.mean(al == 0.0), 100 * np.mean((al != 0.0) & (al != 1.0))))
# parity of the two duration paths on 8192 trajectories, mixed desired durations
nb = 8192
sub = [t[:nb] if t.shape[0] == B else t for t in dv]
des = torch.from_numpy(np.random.default_rng(6).uniform(0.5, 6.0, nb)).to(dev)
a = tb.solve_desired_duration_batch(*sub, des, variant=2)
b = tb.solve_desired_duration_batch(*sub, des, variant=3)
bad = []
for k in ("K", "sd2", "sd", "u", "alpha", "status"):
    x, y = a[k].cpu().numpy().astype(float), b[k].cpu().numpy().astype(float)
    same = (x == y) | (np.isnan(x) & np.isnan(y))
    if not same.all():
        bad.append("%s:%d" % (k, int((~same).sum())))
print("family 3 vs family 2 on %d trajectories:" % nb, "bit-identical" if not bad else "DIFFER " + " ".join(bad))

"""Kernel times of the certified lane kernels per dof (solve, feasible sets, TOPPRAsd) for the library in TOPPRA_HIP_LIB."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
dofs = [int(a) for a in sys.argv[1:]] or [7, 8, 9, 10, 11, 12, 13]
tag = os.path.basename(os.environ.get("TOPPRA_HIP_LIB", "product"))
for d in dofs:
    data = tb.make_synthetic_batch(65536, d, 200)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    out = tb.solve_batch(*dv, variant=3); torch.cuda.synchronize()
    ms = tb.solve_batch_timed(*dv, out, reps=5, variant=3)
    def wall(fn):
        fn(); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); [fn() for _ in range(3)]; ev1.record(); torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / 3
    fs = wall(lambda: tb.feasible_sets_batch(*dv, variant=3))
    sd = wall(lambda: tb.solve_desired_duration_batch(*dv, 3.0, variant=3))
    print("%-22s d %2d  solve %.3f ms  feasible sets %.3f ms  TOPPRAsd %.3f ms" % (tag, d, ms, fs, sd), flush=True)

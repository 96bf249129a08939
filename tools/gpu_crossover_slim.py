"""Family 2 (rows across lanes) against family 3 (certified lane kernel, one fixed-latency round up to 65536 trajectories) at
9..14 dof over batch sizes: where `cert_auto_from(d)` (tpr_kernels.hip) should switch.  One line per dof."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb, _capi
_capi.init(0)
dev = torch.device("cuda", 0)
for d in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else range(9, 16):
    row = []
    for B in (8192, 12288, 16384, 20480, 24576, 28672, 32768, 40960, 49152):
        data = tb.make_synthetic_batch(B, d, 200)
        dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
        t = {}
        for v in (2, 3):
            out = tb.solve_batch(*dv, variant=v); torch.cuda.synchronize()
            t[v] = min(tb.solve_batch_timed(*dv, out, reps=3, variant=v) for _ in range(2))
        row.append("%d: %.2f / %.2f" % (B, t[2], t[3]))
    print("d %2d  (B: family 2 / family 3 ms)  " % d + "   ".join(row), flush=True)

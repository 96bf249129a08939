import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
_capi.init(0)
dev = torch.device("cuda", 0)
row = []
for B in (256, 1024, 2048, 4096, 8192):
    data = tb.make_synthetic_batch(B, 7, 200)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    out = tb.solve_batch(*dv, variant=4); torch.cuda.synchronize()
    row.append("%d: %.3f" % (B, min(tb.solve_batch_timed(*dv, out, reps=5, variant=4) for _ in range(3))))
print("family 4 (B: ms)  " + "   ".join(row))

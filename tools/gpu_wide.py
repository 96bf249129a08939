import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
for B in (4096, 1024, 256):
    data = tb.make_synthetic_batch(B, 7, 200)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    ref = tb.solve_batch(*dv, variant=2)
    for L in ("16", "32", "64"):
        os.environ["TPR_LANES"] = L
        out = tb.solve_batch(*dv, variant=2); torch.cuda.synchronize()
        same = all(bool(torch.equal(torch.nan_to_num(out[k], nan=-7.0), torch.nan_to_num(ref[k], nan=-7.0))) for k in ("sd2", "u", "K"))
        print("B", B, "L", L, "%.3f ms" % tb.solve_batch_timed(*dv, out, reps=5, variant=2), "same bits", same)

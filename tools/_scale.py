import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
for B in (65536, 131072, 262144):
    data = batch.make_synthetic_batch(B, 7, 200)
    dev = [torch.as_tensor(data[k], device="cuda") for k in ("coef", "breaks", "grid", "vlim", "alim")]
    r = {}
    for w in ("1", "2"):
        os.environ["TPR_CERT_WAVES"] = w
        out = batch.solve_batch(*dev, variant=3)
        torch.cuda.synchronize()
        r[w] = batch.solve_batch_timed(*dev, out, 5, variant=3)
    print("B=%d: family3 waves=1 %.3f ms (%.2f M/s)  waves=2 %.3f ms (%.2f M/s)" % (B, r["1"], B / r["1"] / 1e3, r["2"], B / r["2"] / 1e3), flush=True)

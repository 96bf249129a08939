import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
for B in (32768, 65536, 131072, 196608, 262144):
    data = batch.make_synthetic_batch(B, 7, 200)
    dev = [torch.as_tensor(data[k], device="cuda") for k in ("coef", "breaks", "grid", "vlim", "alim")]
    r = {}
    for v in (2, 3):
        out = batch.solve_batch(*dev, variant=v)
        torch.cuda.synchronize()
        r[v] = batch.solve_batch_timed(*dev, out, 5, variant=v)
    print("B=%d: family2 %.3f ms (%.2f M/s)  family3 %.3f ms (%.2f M/s)" % (B, r[2], B / r[2] / 1e3, r[3], B / r[3] / 1e3), flush=True)

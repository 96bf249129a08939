import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
B = 65536
data = batch.make_synthetic_batch(B, 7, 200)
dev = [torch.as_tensor(data[k], device="cuda") for k in ("coef", "breaks", "grid", "vlim", "alim")]
for v in (2, 3):
    out = batch.solve_batch(*dev, variant=v)
    torch.cuda.synchronize()
    print("variant", v, "%.3f ms" % batch.solve_batch_timed(*dev, out, 5, variant=v), flush=True)

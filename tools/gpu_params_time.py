import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from toppra_amd import batch as tb
B, d, N = 65536, 7, 200
data = tb.make_synthetic_batch(B, d, N)
dev = torch.device("cuda", 0)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
for _ in range(2): r = tb.constraint_params_batch(*dv)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = tb.constraint_params_batch(*dv)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
nbytes = sum(v.numel() * 8 for v in r.values())
print("constraint_params 65536x7x200: %.3f ms per call (incl. allocating %.2f GB of outputs), %.0f GB/s" % (ms, nbytes / 1e9, nbytes / ms / 1e6))

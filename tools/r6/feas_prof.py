import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb, _capi
_capi.init(0)
dev = torch.device("cuda", 0)
data = tb.make_synthetic_batch(65536, 7, 200)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
for _ in range(12): tb.feasible_sets_batch(*dv, variant=3)
torch.cuda.synchronize()

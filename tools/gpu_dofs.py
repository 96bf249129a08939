import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
for (B,d,N) in [(65536,8,200),(65536,7,200),(65536,6,500),(65536,4,200)]:
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    out = tb.solve_batch(*dv); torch.cuda.synchronize()
    print(os.environ.get("TOPPRA_HIP_LIB","product")[-14:], B, d, N, "%.3f ms" % tb.solve_batch_timed(*dv, out, reps=3))

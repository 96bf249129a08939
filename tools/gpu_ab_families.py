import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from toppra_amd import batch as tb
for (B,d,N,variant) in [(65536,7,200,0),(65536,7,200,2),(4096,7,200,0)]:
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).cuda() for k in ("coef", "breaks", "grid", "vlim", "alim")]
    out = tb.solve_batch(*dv, variant=variant); torch.cuda.synchronize()
    ms = min(tb.solve_batch_timed(*dv, out, reps=10, variant=variant) for _ in range(3))
    print(os.environ.get("TOPPRA_HIP_LIB","product")[-18:], B, d, N, "variant", variant, "%.3f ms" % ms)

#!/usr/bin/env python
"""Family 2 timings (default certified answers, strict full iteration, Collocation, config 2) for a library build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb
dev = torch.device("cuda", 0)
def run(B, d, N, **kw):
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    out = tb.solve_batch(*dv, **kw); torch.cuda.synchronize()
    return tb.solve_batch_timed(*dv, out, reps=3, **kw)
print(os.environ.get("TOPPRA_HIP_LIB", "product"),
      "f2 %.3f" % run(65536, 7, 200, variant=2), "strict %.3f" % run(65536, 7, 200, strict=True),
      "colloc %.3f" % run(65536, 7, 200, interpolation=False), "c2 %.3f" % run(4096, 7, 200),
      "d12 %.3f" % run(16384, 12, 100), "strict_d6N500 %.3f" % run(65536, 6, 500, strict=True))

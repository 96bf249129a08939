import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
B, d, N = 65536, 7, 200
data = batch.make_synthetic_batch(B, d, N)
out = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=3)
t = out["u"][:, :12]
names = ["bwd loop overhead+K store", "bwd eval+box(+norms)", "batch: publish+ballot", "lane-level checks (upper+lower)", "batch: build rows", "batch results+state update", "batch: pick lanes",
         "fwd overhead(prefetch, update, stores)", "fwd eval", "fwd lp1d", "batch: walk (predict_upper_lp)", "batch: seidel"]
m = t.mean(0)
print("cycles per wave (mean over lanes), share:")
for n, v in zip(names, m):
    print("  %-40s %12.0f  %5.1f%%   per stage %8.0f" % (n, v, 100 * v / m.sum(), v / N))
print("total", m.sum())

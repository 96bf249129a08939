"""Round-5 debugging aid: TOPPRAsd fixture through variants 2 and 3, where do they differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import golden
from toppra_amd import batch, _capi
_capi.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "sd_batch_d5_N80"
fx = golden(name)
args = (fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["desired"], fx["sd_start"], fx["sd_end"])
for rep in range(3):
    a = batch.solve_desired_duration_batch(*args, variant=2)
    b = batch.solve_desired_duration_batch(*args, variant=3)
    for k in ("K", "sd2", "sd", "u", "alpha", "status"):
        x, y = np.asarray(a[k], dtype=float), np.asarray(b[k], dtype=float)
        same = (x == y) | (np.isnan(x) & np.isnan(y))
        if not same.all():
            idx = np.argwhere(~same)
            print(rep, k, "differs at", len(idx), "places; first", idx[:6].tolist(), "v2", x[tuple(idx[0])], "v3", y[tuple(idx[0])])
    print(rep, "status v2", a["status"].tolist()[:40])
    print(rep, "status v3", b["status"].tolist()[:40])
print("B", fx["coef"].shape)

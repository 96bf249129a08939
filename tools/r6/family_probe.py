"""Round-6 forensics: every kernel family's fused solve (variant 1, 2, 4, 5 and, for d <= 8, 3) against the CPU restatement
on a small batch; ONE line per family (which outputs differ, how many trajectories)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toppra_amd import batch, _capi
from oracle import oracle
_capi.init(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 5
data = batch.make_synthetic_batch(128, d, 60, seed=11)
args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
ref = oracle.solve_batch(*args)
for v in (1, 2, 3, 4, 5):
    try:
        out = batch.solve_batch(*args, variant=v)
    except Exception as e:  # a family that does not serve this shape
        print("family", v, "n/a:", str(e)[:80]); continue
    bad = []
    for k in ("K", "sd2", "u", "status"):
        a, b = np.asarray(out[k]), np.asarray(ref[k])
        ne = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else (a != b)
        n = int(ne.reshape(len(a), -1).any(axis=1).sum())
        if n: bad.append("%s:%d" % (k, n))
    print("family", v, "PASS" if not bad else "FAIL " + " ".join(bad))

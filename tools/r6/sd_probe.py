"""One line: does family 3's TOPPRAsd at this dof agree with the rows-across-lanes kernels (variant 2) bit for bit?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import golden
from toppra_amd import batch, _capi
_capi.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "sd_batch_d5_N80"
fx = golden(name)
args = (fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["desired"], fx["sd_start"], fx["sd_end"])
a = batch.solve_desired_duration_batch(*args, variant=2)
b = batch.solve_desired_duration_batch(*args, variant=3)
bad = []
for k in ("K", "sd2", "sd", "u", "alpha", "status"):
    x, y = np.asarray(a[k], dtype=float), np.asarray(b[k], dtype=float)
    same = (x == y) | (np.isnan(x) & np.isnan(y))
    if not same.all():
        bad.append("%s:%d" % (k, int((~same).sum())))
print("PASS" if not bad else "FAIL " + " ".join(bad))

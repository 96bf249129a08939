"""Round-6 forensics: the fastest / slowest TOPPRAsd profile alone (desired duration tiny / huge), variant 2 against 3, raw numbers."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import golden
from toppra_amd import batch, _capi
_capi.init(0)
np.set_printoptions(precision=17, linewidth=200)
fx = golden("sd_batch_d5_N80")
for tag, des in (("fast", 1e-3), ("slow", 1e3)):
    args = (fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], np.full(24, des), fx["sd_start"], fx["sd_end"])
    a = batch.solve_desired_duration_batch(*args, variant=2)
    b = batch.solve_desired_duration_batch(*args, variant=3)
    for t in (0, 1, 2):
        print(tag, "traj", t, "alpha", a["alpha"][t], b["alpha"][t], "sd_start^2", fx["sd_start"][t] ** 2)
        print("  K lo v2", a["K"][t, :6, 0]); print("  K lo v3", b["K"][t, :6, 0])
        print("  K hi v2", a["K"][t, :6, 1]); print("  K hi v3", b["K"][t, :6, 1])
        print("  sd2 v2", a["sd2"][t, :8]); print("  sd2 v3", b["sd2"][t, :8])
        print("  u   v2", a["u"][t, :8]); print("  u   v3", b["u"][t, :8])
        d = np.flatnonzero(~((a["sd2"][t] == b["sd2"][t]) | (np.isnan(a["sd2"][t]) & np.isnan(b["sd2"][t]))))
        print("  sd2 differs at", d[:20].tolist(), "... total", len(d))
        d = np.flatnonzero(~((a["u"][t] == b["u"][t]) | (np.isnan(a["u"][t]) & np.isnan(b["u"][t]))))
        print("  u differs at", d[:20].tolist(), "... total", len(d))

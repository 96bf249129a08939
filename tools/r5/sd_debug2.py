"""Round-5 debugging aid: which shapes trigger the <5 dof, TOPPRAsd> defect (variant 3 against variant 2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import golden
from toppra_amd import batch, _capi
_capi.init(0)
fx = golden("sd_batch_d5_N80")
def run(tag, idx, desired=None, sd0=None, sd1=None, grid=None):
    args = [fx["coef"][idx], fx["breaks"], fx["grid"] if grid is None else grid, fx["vlim"][idx], fx["alim"][idx],
            fx["desired"][idx] if desired is None else desired, fx["sd_start"][idx] if sd0 is None else sd0, fx["sd_end"][idx] if sd1 is None else sd1]
    a = batch.solve_desired_duration_batch(*args, variant=2)
    b = batch.solve_desired_duration_batch(*args, variant=3)
    bad = [k for k in ("K", "sd2", "u", "alpha", "status") if not np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True)]
    nt = int((~((a["sd2"] == b["sd2"]) | (np.isnan(a["sd2"]) & np.isnan(b["sd2"]))).all(axis=1)).sum())
    print("%-40s B %4d differing keys %s, trajectories with differing sd2: %d" % (tag, len(idx), bad, nt), flush=True)
all24 = np.arange(24)
run("fixture as is", all24)
run("tiled to 64", np.resize(all24, 64))
run("tiled to 256", np.resize(all24, 256))
run("first 8", all24[:8])
run("zero boundary velocities", all24, sd0=np.zeros(24), sd1=np.zeros(24))
run("desired huge (alpha 0)", all24, desired=np.full(24, 1e3))
run("desired tiny (alpha 1)", all24, desired=np.full(24, 1e-3))
from toppra_amd import batch as tb
d = tb.make_synthetic_batch(64, 5, 80, seed=3)
for nm, dd in (("synthetic 64 x 5 x 80", d),):
    a = tb.solve_desired_duration_batch(dd["coef"], dd["breaks"], dd["grid"], dd["vlim"], dd["alim"], 2.0, variant=2)
    b = tb.solve_desired_duration_batch(dd["coef"], dd["breaks"], dd["grid"], dd["vlim"], dd["alim"], 2.0, variant=3)
    print(nm, [k for k in ("K", "sd2", "u", "alpha", "status") if not np.array_equal(a[k], b[k], equal_nan=True)])
print("scheme / flags of the fixture:", {k: fx[k] for k in fx if fx[k].ndim == 0})

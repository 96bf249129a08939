"""Round-3 library (commit fc4e883) with its SOUND instantiations of family 3 enabled above 8 dof: does the failure of
round 3 (cert_feasible_kernel<11 | 13, sound> stores nothing / runs 7 x longer) reproduce, and under which variants?"""
import os, sys, time
sys.path.insert(0, os.path.abspath(sys.argv[1]))
import numpy as np
from toppra_amd import batch
print("library:", os.path.abspath(sys.argv[1]), flush=True)

def differ(a, b):
    return int((~((a == b) | (np.isnan(a) & np.isnan(b))).reshape(len(a), -1).all(axis=1)).sum())

for d in (9, 10, 11, 12, 13):
    for B, N in ((1, 2), (1, 40), (2048, 40)):
        data = batch.make_synthetic_batch(B, d, N, seed=60 + d)
        args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        Xf = batch.feasible_sets_batch(*args, True, variant=2, strict=True)
        out = {}
        for sound in (False, True):
            batch.feasible_sets_batch(*args, True, variant=3, sound=sound)
            t0 = time.perf_counter()
            X = batch.feasible_sets_batch(*args, True, variant=3, sound=sound)
            out[sound] = (X, (time.perf_counter() - t0) * 1e3)
        full = batch.solve_batch(*args, variant=2, strict=True)
        s_fast = batch.solve_batch(*args, variant=3)
        s_sound = batch.solve_batch(*args, variant=3, sound=True)
        ds = lambda g: sum(differ(np.asarray(g[k], dtype=float), np.asarray(full[k], dtype=float)) for k in ("K", "sd2", "u"))
        print("d %2d B %5d N %3d: feasible sets fast %d differ (%.2f ms)  SOUND %d differ (%.2f ms)%s | solve fast %d  SOUND %d" % (
            d, B, N, differ(out[False][0], Xf), out[False][1], differ(out[True][0], Xf), out[True][1],
            "  [first row: %s]" % np.array2string(out[True][0][0, :2].ravel(), precision=4) if B == 1 else "", ds(s_fast), ds(s_sound)), flush=True)

#!/usr/bin/env python
"""Robust (conic) path: the rows-across-lanes kernel (8 / 16 lanes per trajectory; Interpolation and Collocation, with
and without feasible sets, up to 16 dof) against the generic lane kernel (variant=1) bit for bit, and timings of BASELINE
config 4.

  python tools/gpu_robust_check.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

bad_total = 0
checks = 0
ELL = [1e-3, 5e-2, 9e-3]


def check(label, got, want):
    global bad_total, checks
    checks += 1
    bad = []
    for k in want:
        x, y = np.asarray(got[k]), np.asarray(want[k])
        eq = (x == y) | (np.isnan(x.astype(float)) & np.isnan(y.astype(float)))
        if not eq.all():
            rows = ~eq.reshape(len(x), -1).all(axis=1)
            bad.append("%s: %d trajectories (first %d), max dev %g" % (k, int(rows.sum()), int(np.flatnonzero(rows)[0]),
                                                                     float(np.nanmax(np.abs(np.nan_to_num(x.astype(float) - y.astype(float)))))))
    if bad:
        bad_total += 1
        print("MISMATCH %-56s %s" % (label, "; ".join(bad)), flush=True)
    else:
        print("ok       %-56s (ok %.2f)" % (label, float((np.asarray(want["status"]) == 0).mean())), flush=True)


def main():
    shapes = [(2048, 7, 60), (40000, 7, 12), (300, 6, 50), (257, 1, 30), (200, 2, 33), (300, 3, 40), (256, 4, 40), (256, 5, 41),
              (256, 8, 40), (128, 9, 30), (128, 12, 30), (96, 16, 20), (65, 7, 1), (3, 7, 2)]
    for B, d, N in shapes:
        data = tb.make_synthetic_batch(B, d, N, seed=500 + d + N)
        rng = np.random.default_rng(d * 17 + N)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
        base = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], ELL)
        cases = [("interp", base, dict()),
                 ("interp + X", base, dict(want_X=True)),
                 ("collocation + X, sd_end", base, dict(interpolation=False, want_X=True, sd_end=sd1)),
                 ("no velocity", (data["coef"], data["breaks"], data["grid"], None, data["alim"], ELL), dict(want_X=True)),
                 ("zero ellipsoid", base[:5] + ([0.0, 0.0, 0.0],), dict())]
        for name, args, kw in cases:
            if B > 10000 and name != "interp":
                continue
            want = tb.robust_solve_batch(*args, variant=1, **kw)
            got = tb.robust_solve_batch(*args, **kw)
            check("B%d d%d N%d %s" % (B, d, N, name), got, want)
    dev = torch.device("cuda", 0)
    d4 = tb.make_synthetic_batch(16384, 7, 100)
    dv4 = [torch.from_numpy(np.ascontiguousarray(d4[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    for label, kw in (("auto", {}), ("auto + X", dict(want_X=True)), ("collocation", dict(interpolation=False)), ("lane kernel", dict(variant=1))):
        tb.robust_solve_batch(*dv4, ELL, **kw)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            tb.robust_solve_batch(*dv4, ELL, **kw)
        ev1.record()
        torch.cuda.synchronize()
        print("time     config 4 (16384 x 7 x 100) %-12s %.3f ms per call" % (label, ev0.elapsed_time(ev1) / 3), flush=True)
    print("checks %d, mismatching %d" % (checks, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

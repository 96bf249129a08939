#!/usr/bin/env python
"""TOPPRAsd (desired duration): the fused family-3 launch (backward scan + fastest / slowest profiles, variant=3) against
the rows-across-lanes path (variant=2) bit for bit -- every output incl. alpha -- on batches of every dof, scaled paths,
boundary velocities, Collocation and desired durations on both sides of the reachable range; the reference's fixtures;
timings at the headline shape.

  python tools/gpu_sd_check.py
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

KEYS = ("K", "sd2", "sd", "u", "status", "alpha")
bad_total = 0
checks = 0


def check(label, got, want, keys=KEYS):
    global bad_total, checks
    checks += 1
    bad = []
    for k in keys:
        x, y = np.asarray(got[k]), np.asarray(want[k])
        eq = (x == y) | (np.isnan(x.astype(float)) & np.isnan(y.astype(float)))
        if not eq.all():
            rows = ~eq.reshape(len(x), -1).all(axis=1)
            bad.append("%s: %d trajectories (first %d), max dev %g" % (k, int(rows.sum()), int(np.flatnonzero(rows)[0]),
                                                                     float(np.nanmax(np.abs(np.nan_to_num(x.astype(float) - y.astype(float)))))))
    if bad:
        bad_total += 1
        print("MISMATCH %-52s %s" % (label, "; ".join(bad)), flush=True)
    else:
        st = np.asarray(want["status"])
        al = np.asarray(got["alpha"])
        print("ok       %-52s (ok %.2f, bisected %.2f)" % (label, float((st == 0).mean()), float(((al > 0) & (al < 1)).mean())), flush=True)


def main():
    shapes = [(4096, 7, 200), (1000, 6, 120), (257, 1, 40), (200, 2, 33), (300, 3, 60), (256, 4, 70), (256, 5, 101), (256, 8, 64),
              (65, 7, 1), (3, 7, 2), (64, 7, 9), (63, 7, 8), (130, 5, 17)]
    for B, d, N in shapes:
        data = tb.make_synthetic_batch(B, d, N, seed=700 + d + N)
        rng = np.random.default_rng(d * 13 + N)
        scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
        sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
        desired = rng.uniform(0.3, 6.0, size=B)
        cases = [("plain", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired), {}),
                 ("boundary", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired), dict(sd_start=sd0, sd_end=sd1)),
                 ("scaled", (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], desired), {}),
                 ("collocation", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired), dict(interpolation=False)),
                 ("acc_only", (data["coef"], data["breaks"], data["grid"], None, data["alim"], desired), {})]
        for name, args, kw in cases:
            want = tb.solve_desired_duration_batch(*args, variant=2, **kw)
            got = tb.solve_desired_duration_batch(*args, variant=3, **kw)
            check("B%d d%d N%d %-11s v3 vs v2" % (B, d, N, name), got, want)
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "sd_batch_*.npz"))):
        fx = dict(np.load(path))
        if fx["coef"].shape[3] > 8:
            continue
        got = tb.solve_desired_duration_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["desired"],
                                              fx["sd_start"], fx["sd_end"], variant=3)
        check("reference fixture %s" % os.path.basename(path)[:-4], got, fx, keys=("K", "sd", "u", "status"))
    dev = torch.device("cuda", 0)
    B, d, N = 65536, 7, 200
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    for desired in (3.0, 1.2):
        ref = tb.solve_desired_duration_batch(*dv, desired, variant=2)
        for variant in (3, 2, 0):
            got = tb.solve_desired_duration_batch(*dv, desired, variant=variant)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                tb.solve_desired_duration_batch(*dv, desired, variant=variant)
            ev1.record()
            torch.cuda.synchronize()
            same = all(bool(torch.equal(torch.nan_to_num(got[k].double(), nan=-7.0), torch.nan_to_num(ref[k].double(), nan=-7.0))) for k in KEYS)
            al = ref["alpha"]
            print("time     65536 x 7 x 200 desired %.1f variant %d: %.3f ms per call, identical to variant 2: %s (bisected %.2f)"
                  % (desired, variant, ev0.elapsed_time(ev1) / 5, same, float(((al > 0) & (al < 1)).double().mean())), flush=True)
            if not same:
                global bad_total
                bad_total += 1
    print("checks %d, mismatching %d" % (checks, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Family 3 (the certified lane kernels) above 8 dof: solve, feasible sets and TOPPRAsd for d = 9..13 against the
rows-across-lanes kernels (full iteration where there is a strict mode) bit for bit, and timings at 65536 x d x 200.

  python tools/gpu_cert_dofs_check.py [--quick]
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


def check(label, got, want, keys):
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
        print("MISMATCH %-56s %s" % (label, "; ".join(bad)), flush=True)
    else:
        print("ok       %s" % label, flush=True)


def main():
    quick = "--quick" in sys.argv
    dofs = (9, 15) if quick else range(9, 16)   # (family 3 is instantiated up to 15 dof)
    for d in dofs:
        B, N = 1500, 60
        data = tb.make_synthetic_batch(B, d, N, seed=800 + d)
        rng = np.random.default_rng(d)
        scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
        sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
        base = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        cases = [("plain", base, {}), ("boundary", base, dict(sd_start=sd0, sd_end=sd1)),
                 ("scaled", (data["coef"] * scale,) + base[1:], {}), ("collocation", base, dict(interpolation=False, sd_end=sd1)),
                 ("acc_only", (data["coef"], data["breaks"], data["grid"], None, data["alim"]), {})]
        for name, args, kw in cases:
            full = tb.solve_batch(*args, variant=2, strict=True, **kw)
            for sound in (False, True):
                got = tb.solve_batch(*args, variant=0 if sound else 3, sound=sound, **kw)
                check("d%d %-11s solve %s vs full iteration (ok %.2f)" % (d, name, "auto, sound certificates" if sound else "v3", float((full["status"] == 0).mean())),
                      got, full, ("K", "sd2", "u", "status"))
            fkw = {k: v for k, v in kw.items() if k == "interpolation"}
            Xf = tb.feasible_sets_batch(*args, variant=2, strict=True, **fkw)
            check("d%d %-11s feasible sets v3 vs full iteration" % (d, name), {"X": tb.feasible_sets_batch(*args, variant=3, **fkw)}, {"X": Xf}, ("X",))
            if args[3] is not None:
                desired = rng.uniform(0.3, 6.0, size=B)
                want = tb.solve_desired_duration_batch(*args, desired, variant=2, **kw)
                got = tb.solve_desired_duration_batch(*args, desired, variant=3, **kw)
                check("d%d %-11s TOPPRAsd v3 vs v2" % (d, name), got, want, ("K", "sd2", "sd", "u", "status", "alpha"))
    dev = torch.device("cuda", 0)
    for d in ((9, 15) if quick else (8, 9, 10, 11, 12, 13, 14, 15)):
        data = tb.make_synthetic_batch(65536, d, 200)
        dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
        out = tb.solve_batch(*dv, variant=2)
        ms2 = tb.solve_batch_timed(*dv, out, 3, variant=2)
        out3 = tb.solve_batch(*dv, variant=3)
        ms3 = tb.solve_batch_timed(*dv, out3, 3, variant=3)
        same = all(bool(torch.equal(torch.nan_to_num(out[k].double(), nan=-7.0), torch.nan_to_num(out3[k].double(), nan=-7.0))) for k in ("K", "sd2", "u", "status"))
        print("time     65536 x %2d x 200: family 3 %.3f ms, family 2 %.3f ms, identical %s" % (d, ms3, ms2, same), flush=True)
        del dv, out, out3
    # above family 3's range: rows across 16 lanes (family 2) up to 16 dof, one trajectory per wave (family 4) up to 32
    for d, B in (() if quick else ((16, 65536), (24, 16384), (32, 16384))):
        data = tb.make_synthetic_batch(B, d, 200)
        dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
        out = tb.solve_batch(*dv)
        ms = tb.solve_batch_timed(*dv, out, 3)
        print("time     %5d x %2d x 200: auto (family %s) %.3f ms, ok fraction %.3f" % (B, d, "2" if d <= 16 else "4", ms,
                                                                               float((out["status"] == 0).double().mean())), flush=True)
        del dv, out
    print("checks %d, mismatching %d" % (checks, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

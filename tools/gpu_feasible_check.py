#!/usr/bin/env python
"""compute_feasible_sets on the certified lane kernel (family 3, variant=3) against the reference's full Seidel iteration
(rows-across-lanes kernel, strict) bit for bit: every dof it serves, scaled paths, Collocation, no velocity constraint,
per-trajectory grids, partial blocks; then the reference's own X fixtures and the timings at the headline shape.

  python tools/gpu_feasible_check.py [--quick]
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

bad_total = 0
checks = 0


def check(label, got, want):
    global bad_total, checks
    checks += 1
    got, want = np.asarray(got), np.asarray(want)
    eq = (got == want) | (np.isnan(got) & np.isnan(want))
    rows = ~eq.reshape(len(got), -1).all(axis=1)
    n = int(rows.sum())
    if n:
        bad_total += 1
        first = int(np.flatnonzero(rows)[0])
        st = np.flatnonzero(~eq[first].reshape(-1, 2).all(axis=1))
        dev = float(np.nanmax(np.abs(np.nan_to_num(got - want))))
        print("MISMATCH %-58s %d trajectories differ; first %d at stages %s: got %s want %s; max |dev| %g"
              % (label, n, first, st[:4].tolist(), got[first, st[0]].tolist(), want[first, st[0]].tolist(), dev), flush=True)
    else:
        print("ok       %-58s (nan fraction %.3f)" % (label, float(np.isnan(want).mean())), flush=True)


def main():
    quick = "--quick" in sys.argv
    shapes = [(4096, 7, 200), (1000, 6, 120), (257, 1, 40), (200, 2, 33), (300, 3, 60), (256, 4, 70), (256, 5, 101), (256, 8, 64),
              (65, 7, 1), (3, 7, 2), (64, 7, 3), (63, 7, 4), (130, 5, 5), (100, 6, 7)]
    if quick:
        shapes = shapes[:4]
    for B, d, N in shapes:
        data = tb.make_synthetic_batch(B, d, N, seed=300 + d + N)
        rng = np.random.default_rng(d * 11 + N)
        scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
        cases = [("plain", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], True)),
                 ("scaled", (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], True)),
                 ("collocation", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], False)),
                 ("scaled colloc.", (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], False)),
                 ("acc_only", (data["coef"], data["breaks"], data["grid"], None, data["alim"], True))]
        # one-sided velocity limits (a positive lower bound on sd)
        v1 = data["vlim"].copy()
        flip = rng.random(B) < 0.5
        v1[flip, 0, 0] = 0.05
        cases.append(("one-sided vlim", (data["coef"], data["breaks"], data["grid"], v1, data["alim"], True)))
        for name, args in cases:
            full = tb.feasible_sets_batch(*args, variant=2, strict=True)
            for sound in (False, True):
                got = tb.feasible_sets_batch(*args, variant=3, sound=sound)
                check("B%d d%d N%d %-14s v3%s vs full iteration" % (B, d, N, name, " sound" if sound else ""), got, full)
            if B <= 1000:
                got4 = tb.feasible_sets_batch(*args, variant=4)
                check("B%d d%d N%d %-14s v4 vs full iteration" % (B, d, N, name), got4, full)
    # per-trajectory grids / breakpoints
    B, d, N = 200, 7, 90
    data = tb.make_synthetic_batch(B, d, N, seed=5)
    rng = np.random.default_rng(5)
    grid_b = np.sort(np.concatenate([np.zeros((B, 1)), rng.random((B, N - 1)), np.ones((B, 1))], axis=1), axis=1)
    grid_b[:, 1:-1] = 0.5 * grid_b[:, 1:-1] + 0.5 * data["grid"][None, 1:-1]
    breaks_b = np.repeat(data["breaks"][None], B, axis=0)
    args = (data["coef"], breaks_b, grid_b, data["vlim"], data["alim"])
    check("per-trajectory grids and breakpoints", tb.feasible_sets_batch(*args, variant=3), tb.feasible_sets_batch(*args, variant=2, strict=True))
    for B, d, N, nway in ((24, 7, 120, 40), (16, 3, 300, 120), (8, 7, 1400, 5)):
        data = tb.make_synthetic_batch(B, d, N, seed=d * 100 + nway, n_waypoints=nway)
        args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        check("long splines / grids B%d d%d N%d nway%d" % (B, d, N, nway), tb.feasible_sets_batch(*args, variant=3),
              tb.feasible_sets_batch(*args, variant=2, strict=True))
    # the reference's own X fixtures
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "batch_*.npz"))):
        fx = dict(np.load(path))
        if "X" not in fx or "coef" not in fx or fx["coef"].ndim != 4 or fx["coef"].shape[3] > 8 or "alim" not in fx:
            continue
        try:
            got = tb.feasible_sets_batch(fx["coef"], fx["breaks"], fx["grid"], fx.get("vlim"), fx.get("alim"),
                                         bool(int(fx["interpolation"])), variant=3)
        except Exception as exc:  # noqa: BLE001
            print("skip     %s: %r" % (os.path.basename(path), exc))
            continue
        check("reference fixture %s" % os.path.basename(path)[:-4], got, fx["X"])
    # timings at the headline shape (device-resident inputs)
    dev = torch.device("cuda", 0)
    B, d, N = 65536, 7, 200
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    ref = tb.feasible_sets_batch(*dv, variant=2, strict=True)
    for variant, kw in ((3, {}), (3, {"sound": True}), (2, {}), (0, {})):
        got = tb.feasible_sets_batch(*dv, variant=variant, **kw)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            tb.feasible_sets_batch(*dv, variant=variant, **kw)
        ev1.record()
        torch.cuda.synchronize()
        same = bool(torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0)))
        print("time     65536 x 7 x 200 variant %d %s: %.3f ms per call, identical to the full iteration: %s"
              % (variant, kw, ev0.elapsed_time(ev1) / 5, same), flush=True)
        if not same:
            global bad_total
            bad_total += 1
    print("checks %d, mismatching %d" % (checks, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

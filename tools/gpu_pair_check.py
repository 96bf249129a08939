#!/usr/bin/env python
"""Kernel family 5 (two trajectories per wave, 32 lanes each) on the GPU box: bit parity against the reference fixtures, against
the full Seidel iteration of family 2 and against family 4 on synthetic batches (1..7 dof, every constraint set, odd batch
sizes, strict mode, per-trajectory grids), and timings around BASELINE config 2.  One line per check; never stops early.

  python tools/gpu_pair_check.py [--quick]
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

KEYS = ("K", "sd2", "u", "status")
bad_total = [0, 0]


def check(label, got, want):
    bad = np.zeros(len(np.asarray(want["status"])), dtype=bool)
    for k in KEYS:
        x, y = np.asarray(got[k]), np.asarray(want[k])
        eq = (x == y) | (np.isnan(x.astype(float)) & np.isnan(y.astype(float)))
        bad |= ~eq.reshape(len(bad), -1).all(axis=1)
    bad_total[1] += 1
    if bad.any():
        bad_total[0] += 1
        dev = max(float(np.nanmax(np.abs(np.nan_to_num(np.asarray(got[k], dtype=float) - np.asarray(want[k], dtype=float))))) for k in ("K", "sd2", "u"))
        print("MISMATCH %-64s %d trajectories differ (first %s), max |dev| %g" % (label, int(bad.sum()), np.flatnonzero(bad)[:6].tolist(), dev), flush=True)
    else:
        print("ok       %s" % label, flush=True)


def fixtures():
    gold = os.path.join(ROOT, "tests", "golden")
    for path in sorted(glob.glob(os.path.join(gold, "batch_*.npz"))):
        fx = dict(np.load(path))
        if fx["coef"].shape[3] > 7:
            continue
        args = (fx["coef"], fx["breaks"], fx["grid"], fx.get("vlim"), fx.get("alim"), fx["sd_start"], fx["sd_end"], bool(int(fx["interpolation"])))
        for kw in (dict(variant=5), dict(variant=5, strict=True)):
            got = tb.solve_batch(*args, want_sd=True, **kw)
            check("fixture %s %s" % (os.path.basename(path)[:-4], kw), {"K": got["K"], "u": got["u"], "status": got["status"], "sd2": got["sd"]},
                  {"K": fx["K"], "u": fx["u"], "status": fx["status"], "sd2": fx["sd"]})
    fx = dict(np.load(os.path.join(gold, "example_kinematics_seed9.npz")))
    for tag in ("n100", "auto"):
        got = tb.solve_batch(fx["coef"], fx["breaks"], fx[tag + "_grid"], fx["vlim"], fx["alim"], want_sd=True, variant=5)
        check("example_kinematics %s" % tag, {"K": got["K"], "u": got["u"], "sd2": got["sd"], "status": got["status"]},
              {"K": fx[tag + "_K"][None], "u": fx[tag + "_u"][None], "sd2": fx[tag + "_sd"][None], "status": np.zeros(1, dtype=np.int32)})


def synthetic(quick):
    shapes = [(512, 7, 200), (301, 6, 120), (257, 1, 40), (200, 2, 33), (300, 3, 60), (255, 4, 70), (256, 5, 101), (1, 7, 100), (3, 7, 1), (5, 2, 2),
              (2, 7, 100), (64, 7, 289)]
    if quick:
        shapes = shapes[:5]
    for B, d, N in shapes:
        data = tb.make_synthetic_batch(B, d, N, seed=900 + d + N)
        rng = np.random.default_rng(d * 7 + N)
        sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
        scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
        tight = data["alim"] * np.where(rng.random((B, 1, 1)) < 0.3, 0.02, 1.0)  # some trajectories nearly uncontrollable
        cases = [("plain", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, None, True)),
                 ("boundary", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1, True)),
                 ("fast_start", (data["coef"], data["breaks"], data["grid"], data["vlim"], tight, 40.0 * sd0, sd1, True)),
                 ("scaled", (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], None, None, True)),
                 ("collocation", (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1, False)),
                 ("acc_only", (data["coef"], data["breaks"], data["grid"], None, data["alim"], None, None, True)),
                 ("vel_only", (data["coef"], data["breaks"], data["grid"], data["vlim"], None, None, None, False))]
        for name, args in cases:
            full = tb.solve_batch(*args, variant=2, strict=True)
            got = tb.solve_batch(*args, variant=5)
            check("B%d d%d N%d %-11s v5 vs full iteration (ok %.2f)" % (B, d, N, name, float((full["status"] == 0).mean())), got, full)
            check("B%d d%d N%d %-11s v5 vs v4" % (B, d, N, name), got, tb.solve_batch(*args, variant=4))
            if name in ("plain", "scaled", "fast_start"):
                check("B%d d%d N%d %-11s v5 strict vs full iteration" % (B, d, N, name), tb.solve_batch(*args, variant=5, strict=True), full)
            got = tb.solve_batch(*args, variant=5, want_sd=True, want_K=False)
            check("B%d d%d N%d %-11s v5 without K, with sd" % (B, d, N, name), dict(got, K=full["K"]), full)
    B, d, N = 201, 7, 90
    data = tb.make_synthetic_batch(B, d, N, seed=5)
    rng = np.random.default_rng(5)
    grid_b = np.sort(np.concatenate([np.zeros((B, 1)), rng.random((B, N - 1)), np.ones((B, 1))], axis=1), axis=1)
    grid_b[:, 1:-1] = 0.5 * grid_b[:, 1:-1] + 0.5 * data["grid"][None, 1:-1]
    breaks_b = np.repeat(data["breaks"][None], B, axis=0)
    args = (data["coef"], breaks_b, grid_b, data["vlim"], data["alim"])
    check("per-trajectory grids and breakpoints", tb.solve_batch(*args, variant=5), tb.solve_batch(*args, variant=2, strict=True))
    for B, d, N, nway in ((24, 7, 120, 40), (17, 3, 300, 120), (9, 7, 700, 5), (6, 5, 64, 200), (4, 6, 40, 400)):
        data = tb.make_synthetic_batch(B, d, N, seed=d * 100 + nway, n_waypoints=nway)
        args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        check("long splines / grids B%d d%d N%d nway%d" % (B, d, N, nway), tb.solve_batch(*args, variant=5), tb.solve_batch(*args, variant=2, strict=True))
    # the sliver family (the reference itself fails on some of these: an LP with an optimum ends "infeasible")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_sliver_hunt as sh
    for B, d, N, seed in ((768, 7, 60, 101), (769, 4, 50, 102)):
        (coef, breaks, grid, vlim, alim, sd0, sd1), _ = sh.family(B, d, N, seed)
        args = (coef, breaks, grid, vlim, alim, sd0, sd1)
        full = tb.solve_batch(*args, variant=2, strict=True)
        check("sliver family B%d d%d N%d (failures %d)" % (B, d, N, int((full["status"] != 0).sum())), tb.solve_batch(*args, variant=5), full)


def big():
    dev = torch.device("cuda", 0)
    for B, d, N, seed in ((4096, 7, 200, 20240924), (8191, 6, 500, 3), (65536, 7, 200, 7)):
        data = tb.make_synthetic_batch(B, d, N, seed=seed)
        dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
        full = tb.solve_batch(*dv, variant=3 if B >= 9216 else 2)
        got = tb.solve_batch(*dv, variant=5)
        check("B%d d%d N%d device tensors v5 vs family %d" % (B, d, N, 3 if B >= 9216 else 2), {k: got[k].cpu().numpy() for k in KEYS},
              {k: full[k].cpu().numpy() for k in KEYS})


def timings():
    dev = torch.device("cuda", 0)

    def kernel_ms(B, d, N, variant, reps=10):
        data = tb.make_synthetic_batch(B, d, N)
        dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
        out = tb.solve_batch(*dv, variant=variant)
        torch.cuda.synchronize()
        return tb.solve_batch_timed(*dv, out, reps=reps, variant=variant)

    for B, d, N in ((1, 7, 100), (2, 7, 100), (64, 7, 200), (512, 7, 200), (1024, 7, 200), (1536, 7, 200), (2048, 7, 200), (3072, 7, 200), (4096, 7, 200),
                    (6144, 7, 200), (8192, 7, 200), (12288, 7, 200), (4096, 3, 200), (4096, 6, 500)):
        row = {}
        for variant in (2, 3, 4, 5, 0):
            if variant == 3 and B < 4096:
                continue
            row["v%d" % variant] = kernel_ms(B, d, N, variant)
        print("timing %6d x %2d x %3d : %s" % (B, d, N, "  ".join("%s %.4f ms" % kv for kv in row.items())), flush=True)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    fixtures()
    synthetic(quick)
    if not quick:
        big()
    timings()
    print("total: %d mismatching of %d checks" % tuple(bad_total), flush=True)

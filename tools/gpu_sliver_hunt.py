#!/usr/bin/env python
"""The concurrent-rows / sliver-pivot family of tests/test_gpu_fullsize.py::test_concurrent_rows_and_sliver_pivots_are_bit_exact
as a hunting tool: fast and sound ("s") modes of every family against the full iteration, with the details of what
differs.
  python tools/gpu_sliver_hunt.py [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch


def family(B, d, N, seed):
    rng = np.random.default_rng(4200 + seed)
    data = batch.make_synthetic_batch(B, d, N, seed=4300 + seed)
    scale = 10.0 ** rng.uniform(-3, 1, size=(B, 1, 1, 1))
    scale[rng.random(B) < 0.5] = 1.0
    coef = data["coef"] * scale
    grid = data["grid"]
    base = batch.solve_batch(coef, data["breaks"], grid, data["vlim"], data["alim"])
    ok = base["status"] == 0
    j = rng.integers(1, N - 1, size=B)
    rows = np.arange(B)
    u0 = np.where(ok, base["u"][rows, j], 0.0)
    x0 = np.where(ok, base["sd2"][rows, j], 0.5)
    off = np.where(rng.random(B) < 0.4, 0.0, 10.0 ** rng.uniform(-8, -2, size=B))
    u0 = u0 + off * rng.standard_normal(B) * np.maximum(1.0, np.abs(u0))
    x0 = np.maximum(x0 + off * rng.standard_normal(B) * np.maximum(1.0, np.abs(x0)), 0.0)
    par = batch.constraint_params_batch(coef, data["breaks"], grid, data["vlim"], data["alim"])
    qs, qss = par["qs"][rows, j], par["qss"][rows, j]
    alim = data["alim"].copy()
    joints = np.argsort(rng.random((B, d)), axis=1)[:, :3]
    for t in range(min(3, d)):
        k = joints[:, t]
        val = qs[rows, k] * u0 + qss[rows, k] * x0
        eps = 10.0 ** rng.uniform(-13, -8, size=B) * rng.choice([-1.0, 1.0], size=B) * np.maximum(1.0, np.abs(val))
        upper = rng.random(B) < 0.5
        width = 10 + 2 * rng.random(B)
        amax = np.where(upper, val + eps, val + eps + width)
        amin = np.where(upper, val + eps - width, val + eps)
        alim[rows, k, 0], alim[rows, k, 1] = amin, amax
    sd1 = np.where(rng.random(B) < 0.3, 0.2 * rng.random(B), 0.0)
    return (coef, data["breaks"], grid, data["vlim"], alim, None, sd1), j


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    total = 0
    for r in range(rounds):
        for B, d, N, seed in ((16384, 7, 60, 1 + 10 * r), (16384, 4, 50, 2 + 10 * r), (16384, 3, 40, 3 + 10 * r), (8192, 8, 48, 4 + 10 * r),
                              (16384, 5, 70, 5 + 10 * r), (16384, 6, 64, 6 + 10 * r), (8192, 2, 40, 7 + 10 * r), (4096, 12, 40, 8 + 10 * r),
                              (8192, 9, 40, 9 + 10 * r), (8192, 10, 36, 10 + 10 * r), (8192, 11, 32, 11 + 10 * r), (8192, 13, 30, 12 + 10 * r)):
            args, j = family(B, d, N, seed)
            full = batch.solve_batch(*args, strict=True)
            total += B
            line = "B %5d d %2d N %3d seed %3d ok %.3f :" % (B, d, N, seed, (full["status"] == 0).mean())
            for variant, sound in ((2, False), (3, False), (2, True), (3, True), (4, False)):
                if variant == 3 and d > 13:
                    continue
                fast = batch.solve_batch(*args, variant=variant, sound=sound)
                bad = np.zeros(B, dtype=bool)
                for k in ("K", "sd2", "u"):
                    eq = (fast[k] == full[k]) | (np.isnan(fast[k]) & np.isnan(full[k]))
                    bad |= ~eq.reshape(B, -1).all(axis=1)
                bad |= fast["status"] != full["status"]
                line += "  v%d%s: %d differ (%d where the reference succeeds)" % (
                    variant, "s" if sound else "", bad.sum(), (bad & (full["status"] == 0)).sum())
                for b in np.flatnonzero(bad)[:3]:
                    Kf, Ks = fast["K"][b], full["K"][b]
                    eq = (Kf == Ks) | (np.isnan(Kf) & np.isnan(Ks))
                    st = np.flatnonzero(~eq.all(axis=1))
                    i = st.max() if len(st) else -1
                    print("    v%d traj %d (engineered stage %d): status %d vs %d; K differs at %d stages, last stage %d: fast %s full %s" % (
                        variant, b, j[b], fast["status"][b], full["status"][b], len(st), i, Kf[i] if i >= 0 else None, Ks[i] if i >= 0 else None))
            print(line, flush=True)
    print("total trajectories", total)

#!/usr/bin/env python
"""Stress of the round-3 kernels against their full-iteration / two-scan counterparts, bit for bit, on the two problem
families of tools/gpu_shortcut_stress.py -- (1) paths scaled by 1e-5 .. 3, where the reference's absolute tolerances bite
and it fails up to half of the time; (2) irregular problems: asymmetric limits incl. positive lower velocity limits,
joints that stand still, non-uniform knots and grids, non-zero boundary velocities, 4-9 waypoints:

  feasible sets   certified lane kernel (variant 3, fast and sound)      vs  rows across lanes, strict (full iteration)
  TOPPRAsd        fused certified launch + wave-per-trajectory finish    vs  rows across lanes for both scans
  solve 9-15 dof  certified lane kernel (fast and sound)                 vs  rows across lanes, strict

  python tools/gpu_r3_stress.py [rounds]
"""
import os
import sys

import numpy as np

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch  # noqa: E402

totals = {}


def differ(a, b, keys):
    B = len(np.asarray(a[keys[0]]))
    bad = np.zeros(B, bool)
    for k in keys:
        x, y = np.asarray(a[k]).reshape(B, -1), np.asarray(b[k]).reshape(B, -1)
        bad |= (~((x == y) | (np.isnan(x.astype(float)) & np.isnan(y.astype(float))))).any(axis=1)
    return bad


def tally(name, B, bad):
    t = totals.setdefault(name, [0, 0])
    t[0] += B
    t[1] += int(bad.sum())


def scaled(seed, B, d, N):
    data = batch.make_synthetic_batch(B, d, N, seed=200 + seed)
    rng = np.random.default_rng(seed)
    scale = (10.0 ** rng.uniform(-5, 0.5, size=B))[:, None, None, None]
    return (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"]), None, None, rng


def irregular(seed, B, d, N, nw):
    rng = np.random.default_rng(3000 + seed)
    knots = np.concatenate([[0.0], np.sort(rng.random(nw - 2)) * 0.9 + 0.05, [1.0]])
    way = rng.standard_normal((B, nw, d))
    still = rng.random((B, d)) < 0.08
    way = np.where(still[:, None, :], way[:, :1, :], way)
    coef, breaks = batch.spline_coefficients(knots, way)
    grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]])
    grid = 0.6 * grid + 0.4 * np.linspace(0, 1, N + 1)
    vhi = 5 + 25 * rng.random((B, d)); vlo = -(5 + 25 * rng.random((B, d)))
    vlo = np.where(rng.random((B, d)) < 0.03, 0.05 * rng.random((B, d)), vlo)
    ahi = 5 + 10 * rng.random((B, d)); alo = -(5 + 10 * rng.random((B, d)))
    vlim = np.ascontiguousarray(np.stack([vlo, vhi], -1)); alim = np.ascontiguousarray(np.stack([alo, ahi], -1))
    sd0 = np.where(rng.random(B) < 0.4, 0.3 * rng.random(B), 0.0)
    sd1 = np.where(rng.random(B) < 0.4, 0.3 * rng.random(B), 0.0)
    return (coef, breaks, grid, vlim, alim), sd0, sd1, rng


for rnd in range(ROUNDS):
    problems = [("scaled", scaled(10 * rnd + s, B, d, N), (B, d, N)) for s, (B, d, N) in
                enumerate([(65536, 7, 200), (65536, 6, 100), (65536, 3, 100), (65536, 8, 64), (65536, 5, 150), (49152, 4, 120)])]
    problems += [("irregular", irregular(10 * rnd + s, B, d, N, nw), (B, d, N)) for s, (B, d, N, nw) in
                 enumerate([(65536, 7, 200, 5), (65536, 6, 120, 9), (40000, 8, 90, 6), (65536, 3, 150, 4), (49152, 5, 64, 7)])]
    for fam, (args, sd0, sd1, rng), (B, d, N) in problems:
        for interp in (True, False):
            full = batch.feasible_sets_batch(*args, interp, variant=2, strict=True)
            for sound in (False, True):
                got = batch.feasible_sets_batch(*args, interp, variant=3, sound=sound)
                tally("feasible sets, %s%s" % (fam, ", sound" if sound else ""), B, differ({"X": got}, {"X": full}, ("X",)))
        if d <= 8:  # (the solve of 9-13 dof has its own loop below)
            full = batch.solve_batch(*args, sd0, sd1, variant=2, strict=True)
            got = batch.solve_batch(*args, sd0, sd1, variant=3, sound=True)
            tally("solve <= 8 dof, %s, sound" % fam, B, differ(got, full, ("K", "sd2", "u", "status")))
        desired = rng.uniform(0.2, 8.0, size=B)
        want = batch.solve_desired_duration_batch(*args, desired, sd0, sd1, variant=2)
        got = batch.solve_desired_duration_batch(*args, desired, sd0, sd1, variant=3)
        bad = differ(got, want, ("K", "sd2", "sd", "u", "status", "alpha"))
        tally("TOPPRAsd, %s" % fam, B, bad)
        print("round %d %-9s B=%d d=%d N=%d: feasible / TOPPRAsd done; TOPPRAsd mismatching %d, bisected %.2f, ok %.2f"
              % (rnd, fam, B, d, N, int(bad.sum()), float(((want["alpha"] > 0) & (want["alpha"] < 1)).mean()), float((want["status"] == 0).mean())), flush=True)
    for s, d in enumerate((9, 10, 11, 12, 13, 14, 15)):
        for fam, (args, sd0, sd1, rng) in (("scaled", scaled(100 + 10 * rnd + s, 32768, d, 100)), ("irregular", irregular(100 + 10 * rnd + s, 32768, d, 80, 6))):
            full = batch.solve_batch(*args, sd0, sd1, variant=2, strict=True)
            for sound in (False, True):
                got = batch.solve_batch(*args, sd0, sd1, variant=0 if sound else 3, sound=sound)
                tally("solve 9-15 dof, %s%s" % (fam, ", sound" if sound else ""), 32768, differ(got, full, ("K", "sd2", "u", "status")))
            Xf = batch.feasible_sets_batch(*args, variant=2, strict=True)
            for sound in (False, True):
                tally("feasible sets 9-15 dof, %s%s" % (fam, ", sound" if sound else ""), 32768,
                      differ({"X": batch.feasible_sets_batch(*args, variant=0 if sound else 3, sound=sound)}, {"X": Xf}, ("X",)))
            desired = rng.uniform(0.2, 8.0, size=32768)
            tally("TOPPRAsd 9-15 dof, %s" % fam, 32768, differ(batch.solve_desired_duration_batch(*args, desired, sd0, sd1, variant=3),
                                                              batch.solve_desired_duration_batch(*args, desired, sd0, sd1, variant=2),
                                                              ("K", "sd2", "sd", "u", "status", "alpha")))
        print("round %d d=%d solve done" % (rnd, d), flush=True)
for name, (n, bad) in sorted(totals.items()):
    print("%-40s %9d trajectories, %d mismatching" % (name, n, bad))
print("total mismatching %d" % sum(b for _, b in totals.values()))

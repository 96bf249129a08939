#!/usr/bin/env python
"""Cross-check of the lower-bound shortcut: the default library vs a build with
-DTPR_PREDICT_LOWER=0 (every lower LP through full Seidel) must give identical bits.
usage: gpu_predict_check.py dump <out.npz>   (run once per library via TOPPRA_HIP_LIB)
       gpu_predict_check.py compare a.npz b.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [(65536, 7, 200, 20240924), (65536, 7, 200, 7), (32768, 6, 500, 3), (16384, 3, 100, 5), (16384, 8, 64, 6),
         (8192, 1, 50, 8), (8192, 5, 300, 9)]

if sys.argv[1] == "dump":
    import torch
    from toppra_amd import batch as tb
    from tests.helpers import golden, batch_fixtures, fixture_problem
    out = {}
    rng = np.random.default_rng(0)
    for B, d, N, seed in CASES:
        data = tb.make_synthetic_batch(B, d, N, seed=seed)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)   # some non-zero end velocities
        sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
        r = tb.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1, variant=2)
        for k in ("K", "sd2", "u", "status"):
            out["%d_%d_%d_%d_%s" % (B, d, N, seed, k)] = r[k]
        # scaled-down / badly conditioned variants
        scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
        r = tb.solve_batch(data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], variant=2)
        for k in ("K", "sd2", "u", "status"):
            out["tiny_%d_%d_%d_%d_%s" % (B, d, N, seed, k)] = r[k]
    for name in batch_fixtures():
        fx = golden(name)
        coef, breaks, grid, vlim, alim, s0, s1, interp = fixture_problem(fx)
        if not interp or coef.shape[3] > 8:
            continue
        r = tb.solve_batch(coef, breaks, grid, vlim, alim, s0, s1, interp, variant=2)
        for k in ("K", "sd2", "u", "status"):
            out["fx_%s_%s" % (name, k)] = r[k]
    np.savez(sys.argv[2], **out)
    print("dumped", len(out), "arrays")
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    nstage = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        if k.endswith("_K"):
            nstage += a[k].shape[0] * (a[k].shape[1] - 1)
        if not same:
            bad += 1
            d = np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float)))
            print("MISMATCH", k, "max dev", d, "count", int(np.sum(~((a[k] == b[k]) | (np.isnan(a[k].astype(float)) & np.isnan(b[k].astype(float)))))))
    print("compared %d arrays, %d mismatching; %.1f M backward stages cross-checked" % (len(a.files), bad, nstage / 1e6))

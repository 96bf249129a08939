#!/usr/bin/env python
"""Stress of the lower-bound shortcut against the full iteration (TPR_STRICT_SEIDEL) on scaled
problems, where the reference's absolute tolerances start to bite.  Prints mismatching
trajectories per scale decade."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch

tot = mism = 0
for seed, (B, d, N) in enumerate([(65536, 7, 200), (65536, 6, 100), (32768, 6, 500), (65536, 3, 100), (65536, 5, 150),
                                  (65536, 8, 64), (65536, 7, 200), (65536, 4, 120)]):
    data = batch.make_synthetic_batch(B, d, N, seed=100 + seed)
    rng = np.random.default_rng(seed)
    logs = rng.uniform(-5, 0.5, size=B)
    scale = (10.0 ** logs)[:, None, None, None]
    args = (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"])
    fast = batch.solve_batch(*args)
    full = batch.solve_batch(*args, strict=True)
    bad = np.zeros(B, bool)
    for k in ("K", "sd2", "u"):
        a, b = fast[k].reshape(B, -1), full[k].reshape(B, -1)
        bad |= (~((a == b) | (np.isnan(a) & np.isnan(b)))).any(axis=1)
    bad |= fast["status"] != full["status"]
    tot += B; mism += int(bad.sum())
    hist = np.histogram(logs[bad], bins=np.arange(-5, 1.5, 0.5))[0]
    okfrac = (full["status"] == 0).mean()
    print("B=%d d=%d N=%d: mismatching trajectories %d (by log10 scale bin from -5: %s), reference-ok fraction %.3f"
          % (B, d, N, bad.sum(), hist.tolist(), okfrac))
print("total %d trajectories, %d mismatching" % (tot, mism))

#!/usr/bin/env python
"""Collocation on the certified lane kernel against the full Seidel iteration (family 2, TPR_STRICT_SEIDEL), bit for bit:
every dof family 3 serves, scaled paths, non-zero boundary velocities, with and without the velocity constraint."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = bad = 0
for r in range(rounds):
    for d in range(1, 9):
        B, N = 32768, 40 + 20 * ((d + r) % 5)
        data = batch.make_synthetic_batch(B, d, N, seed=1000 * r + d, n_waypoints=4 + (d + r) % 4)
        rng = np.random.default_rng(77 * r + d)
        sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
        sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
        scale = 10.0 ** rng.uniform(-5, 0, size=(B, 1, 1, 1))
        vlim = data["vlim"] if (d + r) % 3 else None
        args = (data["coef"] * scale, data["breaks"], data["grid"], vlim, data["alim"], sd0, sd1, False)
        full = batch.solve_batch(*args, strict=True)
        fast = batch.solve_batch(*args, variant=3)
        mism = np.zeros(B, bool)
        for k in ("K", "sd2", "u"):
            mism |= ~np.all((fast[k] == full[k]) | (np.isnan(fast[k]) & np.isnan(full[k])), axis=tuple(range(1, fast[k].ndim)))
        mism |= fast["status"] != full["status"]
        total += B; bad += int(mism.sum())
        print("round %d d=%d N=%d vel=%s: mismatching %d, status counts %s" % (r, d, N, vlim is not None, int(mism.sum()), np.bincount(full["status"], minlength=3).tolist()))
print("collocation total %d trajectories, %d mismatching" % (total, bad))

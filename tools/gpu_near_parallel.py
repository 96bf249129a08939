#!/usr/bin/env python
"""Adversarial inputs for the certificates: joints whose constraint rows are parallel to within 1e-6 .. 1e-14
(one joint is a scaled copy of another, so that their rows (q', q'') point the same way at EVERY gridpoint),
with limits chosen so that either copy can be the binding one.  Default path vs full iteration, bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch

tot = bad_tot = 0
for seed, (B, d, N) in enumerate([(32768, 7, 200), (32768, 4, 100), (16384, 8, 120), (32768, 3, 150)]):
    rng = np.random.default_rng(900 + seed)
    way = rng.standard_normal((B, 5, d))
    eps = 10.0 ** rng.uniform(-14, -6, size=B)
    scale = rng.choice([1.0, -1.0, 0.5, 2.0, 3.0], size=B)
    src, dst = rng.integers(0, d, size=B), rng.integers(0, d, size=B)
    dst = np.where(dst == src, (src + 1) % d, dst)
    rows = np.arange(B)
    way[rows, :, dst] = way[rows, :, src] * (scale * (1 + eps))[:, None]
    twist = rng.random(B) < 0.5                     # half of them: tilt the copy by eps instead of scaling it
    way[rows[twist], :, dst[twist]] += eps[twist, None] * rng.standard_normal((twist.sum(), 5))
    coef, breaks = batch.spline_coefficients(np.linspace(0, 1, 5), way)
    vmax = 10 + 20 * rng.random((B, d)); amax = 10 + 2 * rng.random((B, d))
    # limits of the copy within a few percent of the scaled original's, so that the binding copy alternates
    amax[rows, dst] = amax[rows, src] * np.abs(scale) * (1 + 0.02 * rng.standard_normal(B))
    vmax[rows, dst] = vmax[rows, src] * np.abs(scale) * (1 + 0.02 * rng.standard_normal(B))
    vlim = np.ascontiguousarray(np.stack([-vmax, vmax], -1)); alim = np.ascontiguousarray(np.stack([-amax, amax], -1))
    grid = np.linspace(0, 1, N + 1)
    sd1 = np.where(rng.random(B) < 0.3, 0.2 * rng.random(B), 0.0)
    args = (coef, breaks, grid, vlim, alim, None, sd1)
    full = batch.solve_batch(*args, strict=True)
    for variant in (2, 3):
        fast = batch.solve_batch(*args, variant=variant)
        bad = np.zeros(B, bool)
        for k in ("K", "sd2", "u"):
            a, b = fast[k].reshape(B, -1), full[k].reshape(B, -1)
            bad |= (~((a == b) | (np.isnan(a) & np.isnan(b)))).any(axis=1)
        bad |= fast["status"] != full["status"]
        tot += B; bad_tot += int(bad.sum())
        hist = np.histogram(np.log10(eps[bad]), bins=np.arange(-14, -5, 1))[0] if bad.any() else []
        print("B=%d d=%d N=%d variant %d: mismatching trajectories %d %s; status counts %s" % (
            B, d, N, variant, bad.sum(), list(hist), np.bincount(full["status"], minlength=3).tolist()))
print("near-parallel total %d trajectory solves, %d mismatching" % (tot, bad_tot))

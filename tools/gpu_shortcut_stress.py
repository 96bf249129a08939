#!/usr/bin/env python
"""Stress of the certified answers (default path) against the full iteration (TPR_STRICT_SEIDEL):
(1) problems scaled by 1e-5 .. 3 where the reference's absolute tolerances start to bite (mismatching
trajectories per scale decade), (2) irregular problems: asymmetric limits incl. positive lower velocity
limits, joints that stand still, non-uniform knots and grids, non-zero boundary velocities, 4-9 waypoints."""
import os, sys
import numpy as np
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # repeat everything with fresh seeds
VARIANT = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: the default path; 2: the rows-across-lanes family with its shortcuts
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch

tot = mism = 0
for rnd in range(ROUNDS):
  for seed, (B, d, N) in enumerate([(65536, 7, 200), (65536, 6, 100), (32768, 6, 500), (65536, 3, 100), (65536, 5, 150),
                                    (65536, 8, 64), (65536, 7, 200), (65536, 4, 120)]):
      data = batch.make_synthetic_batch(B, d, N, seed=100 + seed + 1000 * rnd)
      rng = np.random.default_rng(seed + 1000 * rnd)
      logs = rng.uniform(-5, 0.5, size=B)
      scale = (10.0 ** logs)[:, None, None, None]
      args = (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"])
      fast = batch.solve_batch(*args, variant=VARIANT)
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


# ---- (2) irregular problems -------------------------------------------------------------------------
tot2 = mism2 = 0
for rnd in range(ROUNDS):
  for seed, (B, d, N, nw) in enumerate([(65536, 7, 200, 5), (65536, 6, 120, 9), (40000, 8, 90, 6), (65536, 3, 150, 4), (49152, 5, 64, 7)]):
      rng = np.random.default_rng(1000 + seed + 7919 * rnd)
      knots = np.concatenate([[0.0], np.sort(rng.random(nw - 2)) * 0.9 + 0.05, [1.0]])
      way = rng.standard_normal((B, nw, d))
      still = rng.random((B, d)) < 0.08                       # joints that do not move
      way = np.where(still[:, None, :], way[:, :1, :], way)
      coef, breaks = batch.spline_coefficients(knots, way)
      grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]])
      grid = 0.6 * grid + 0.4 * np.linspace(0, 1, N + 1)       # non-uniform, steps bounded away from 0
      vhi = 5 + 25 * rng.random((B, d)); vlo = -(5 + 25 * rng.random((B, d)))
      poslow = rng.random((B, d)) < 0.03                       # a few positive lower velocity limits
      vlo = np.where(poslow, 0.05 * rng.random((B, d)), vlo)
      ahi = 5 + 10 * rng.random((B, d)); alo = -(5 + 10 * rng.random((B, d)))
      vlim = np.ascontiguousarray(np.stack([vlo, vhi], -1)); alim = np.ascontiguousarray(np.stack([alo, ahi], -1))
      sd0 = np.where(rng.random(B) < 0.4, 0.3 * rng.random(B), 0.0)
      sd1 = np.where(rng.random(B) < 0.4, 0.3 * rng.random(B), 0.0)
      args = (coef, breaks, grid, vlim, alim, sd0, sd1)
      fast = batch.solve_batch(*args, variant=VARIANT)
      full = batch.solve_batch(*args, strict=True)
      bad = np.zeros(B, bool)
      for k in ("K", "sd2", "u"):
          a, b = fast[k].reshape(B, -1), full[k].reshape(B, -1)
          bad |= (~((a == b) | (np.isnan(a) & np.isnan(b)))).any(axis=1)
      bad |= fast["status"] != full["status"]
      tot2 += B; mism2 += int(bad.sum())
      print("irregular B=%d d=%d N=%d waypoints=%d: mismatching trajectories %d, status counts (ok/uncontrollable/unknown) %s"
            % (B, d, N, nw, bad.sum(), np.bincount(full["status"], minlength=3).tolist()))
print("irregular total %d trajectories, %d mismatching" % (tot2, mism2))

#!/usr/bin/env python
"""Relaxed mode (TPR_RELAXED_LOWER) vs the reference-generated fixtures and vs exact mode."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toppra_amd import batch as tb
from tests.helpers import golden, batch_fixtures, fixture_problem

for name in batch_fixtures():
    fx = golden(name)
    coef, breaks, grid, vlim, alim, sd0, sd1, interp = fixture_problem(fx)
    if not interp or coef.shape[3] > 8:
        continue
    got = tb.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, interp, want_sd=True, variant=2, relaxed=True)
    ok = fx["status"] == 0
    st = np.array_equal(got["status"], fx["status"])
    devK = np.nanmax(np.abs(got["K"] - fx["K"])) if np.isfinite(fx["K"]).any() else 0
    devx = np.nanmax(np.abs(got["sd"][ok] ** 2 - fx["sd"][ok] ** 2)) if ok.any() else 0
    devu = np.nanmax(np.abs(got["u"][ok] - fx["u"][ok])) if ok.any() else 0
    nanpat = np.array_equal(np.isnan(got["K"]), np.isnan(fx["K"]))
    print("%-32s status_equal=%s nan_pattern_equal=%s max|dK|=%.2e max|d sd^2|=%.2e max|du|=%.2e" % (name, st, nanpat, devK, devx, devu))

d = tb.make_synthetic_batch(65536, 7, 200)
dev = torch.device("cuda", 0)
dv = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")}
ex = tb.solve_batch(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"])
rx = tb.solve_batch(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], relaxed=True)
print("headline: status equal", bool((ex["status"] == rx["status"]).all()), "max|d sd^2| %.2e" % float((ex["sd2"] - rx["sd2"]).abs().max()),
      "max|dK| %.2e" % float((ex["K"] - rx["K"]).abs().max()), "max|du| %.2e" % float((ex["u"] - rx["u"]).abs().max()))
for relaxed in (False, True):
    ms = tb.solve_batch_timed(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], ex, reps=5, relaxed=relaxed)
    print("relaxed=%s kernel_ms %.3f traj/s %.3e" % (relaxed, ms, 65536 / ms * 1e3))

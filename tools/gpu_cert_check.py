#!/usr/bin/env python
"""Family 3 (certified lane kernel) against family 2 in strict mode: bit comparison + kernel time."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch

def same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())

for B, d, N, scaled in [(4096, 7, 200, False), (65536, 7, 200, False), (65536, 6, 500, False), (16384, 3, 100, False),
                        (65536, 7, 200, True)]:
    data = batch.make_synthetic_batch(B, d, N, seed=7 + N)
    coef = data["coef"]
    if scaled:
        rng = np.random.default_rng(1)
        coef = coef * (10.0 ** rng.uniform(-5, 0.5, size=B))[:, None, None, None]
    args = (coef, data["breaks"], data["grid"], data["vlim"], data["alim"])
    ref = batch.solve_batch(*args, variant=2, strict=True)
    new = batch.solve_batch(*args, variant=3)
    ok = all(same(ref[k], new[k]) for k in ("K", "sd2", "u")) and (ref["status"] == new["status"]).all()
    bad = 0
    if not ok:
        for k in ("K", "sd2", "u"):
            a, b = ref[k].reshape(B, -1), new[k].reshape(B, -1)
            bad = max(bad, int((~((a == b) | (np.isnan(a) & np.isnan(b)))).any(axis=1).sum()))
    dev = [torch.as_tensor(x, device="cuda") for x in args]
    times = {}
    for v in (2, 3):
        out = batch.solve_batch(*dev, variant=v)
        torch.cuda.synchronize()
        times[v] = batch.solve_batch_timed(*dev, out, 5, variant=v)
    print("B=%d d=%d N=%d scaled=%s: identical=%s (bad trajectories %d)  kernel ms: family2 %.3f  family3 %.3f"
          % (B, d, N, scaled, ok, bad, times[2], times[3]), flush=True)

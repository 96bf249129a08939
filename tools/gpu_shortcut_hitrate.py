#!/usr/bin/env python
"""How the stage LPs get answered (needs a -DTPR_DEBUG_PREDICT build, which makes the solve kernels
return per-trajectory counters in `status`).
family 3: bits 0-9 lower-bound LPs handed to the batches, 10-19 upper-bound LPs handed to the batches,
          20-25 / 26-31 of those, the ones that needed the full Seidel iteration.
family 2: bits 0-9 lower-bound LPs answered by the shortcut, 10-19 upper-bound LPs answered,
          20-29 stages where every trajectory of the wave was answered."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb
for B, d, N in [(65536, 7, 200), (65536, 6, 500), (16384, 3, 100)]:
    if d != 7 and os.environ.get("TPR_DEV_BUILD"):
        continue
    data = tb.make_synthetic_batch(B, d, N)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    st = tb.solve_batch(*args, variant=3)["status"].astype(np.int64) & 0xffffffff
    lo, up, slo, sup = st & 1023, (st >> 10) & 1023, (st >> 20) & 63, (st >> 26) & 63
    print("B=%d d=%d N=%d family 3: lane-level certificate answers %.3f%% of the upper-bound and %.3f%% of the lower-bound LPs; "
          "batches (walk): %.3f / %.3f LPs per trajectory, of which full iteration: %.3f / %.3f"
          % (B, d, N, 100 - 100.0 * up.mean() / N, 100 - 100.0 * lo.mean() / N, up.mean(), lo.mean(), sup.mean(), slo.mean()))
    w = (up + lo).reshape(-1, 64).sum(1) / N
    print("   LPs handed to the batches per wave and stage: mean %.2f" % w.mean())
    st = tb.solve_batch(*args, variant=2)["status"]
    lo, up, wv = st & 1023, (st >> 10) & 1023, (st >> 20) & 1023
    print("   family 2: lower shortcut %.3f%%  upper shortcut %.3f%%  whole-wave upper %.3f%% of %d stages"
          % (100.0 * lo.mean() / N, 100.0 * up.mean() / N, 100.0 * wv.mean() / N, N))

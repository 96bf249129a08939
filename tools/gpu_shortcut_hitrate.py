#!/usr/bin/env python
"""Hit rates of the certified shortcuts (needs a -DTPR_DEBUG_PREDICT build, which makes the solve
kernel return per-trajectory counters in `status`: bits 0-9 lower-bound LPs answered, bits 10-19
upper-bound LPs answered, bits 20-29 stages where every trajectory of the wave was answered)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb
for B, d, N in [(65536, 7, 200), (65536, 6, 500), (16384, 3, 100)]:
    data = tb.make_synthetic_batch(B, d, N)
    st = tb.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=2)["status"]
    lo, up, wv = st & 1023, (st >> 10) & 1023, (st >> 20) & 1023
    print("B=%d d=%d N=%d: lower shortcut %.3f%%  upper shortcut %.3f%%  whole-wave upper %.3f%% of %d stages"
          % (B, d, N, 100.0 * lo.mean() / N, 100.0 * up.mean() / N, 100.0 * wv.mean() / N, N))

#!/usr/bin/env python
"""Hit rate of the certified lower-bound shortcut (needs a -DTPR_DEBUG_PREDICT build, which makes
the solve kernel return the per-trajectory hit count in `status`)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch as tb
for B, d, N in [(65536, 7, 200), (65536, 6, 500), (16384, 3, 100)]:
    data = tb.make_synthetic_batch(B, d, N)
    st = tb.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=2)["status"]
    print("B=%d d=%d N=%d: shortcut answered %.4f%% of the %d lower-bound LPs per trajectory (min %d, max %d)"
          % (B, d, N, 100.0 * st.mean() / N, N, st.min(), st.max()))

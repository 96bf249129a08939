#!/usr/bin/env python
"""tpr_param_spline_batch at several batch sizes (device tensors, torch events around REPS calls): a batch that fits
the chip in one round of waves gives the latency of one wave, the headline batch the throughput.
  python tools/gpu_param_spline_probe.py [B ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toppra_amd import batch as tb

dev = torch.device("cuda", 0)
d, N = int(os.environ.get("PS_D", 7)), int(os.environ.get("PS_N", 200))
for B in [int(a) for a in sys.argv[1:]] or [2048, 16384, 65536]:
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    sol = tb.solve_batch(*dv, want_sd=True, want_K=False, want_u=False)
    for variant in (0, 1):
        tb.param_spline_batch(dv[0], dv[1], dv[2], sol["sd"], variant=variant)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            tb.param_spline_batch(dv[0], dv[1], dv[2], sol["sd"], variant=variant)
        e1.record()
        torch.cuda.synchronize()
        print("B %6d d %d N %d %s: %.3f ms per call" % (B, d, N, "fused  " if variant == 0 else "generic", e0.elapsed_time(e1) / reps), flush=True)

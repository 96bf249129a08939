#!/usr/bin/env python
"""Where the default kernel (family 3) spends its cycles: per-phase cycle counters of every wave,
accumulated in-kernel with s_memtime by a -DTPR_CERT_TIMING build and returned through `u`.

    python -m toppra_amd.build -DTPR_CERT_TIMING -DTPR_CERT_DEV --out=build_dbg/libtoppra_tim.so   (build container)
    TOPPRA_HIP_LIB=build_dbg/libtoppra_tim.so python tools/gpu_cert_phases.py                     (GPU box)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
B, d, N = 65536, 7, 200
data = batch.make_synthetic_batch(B, d, N)
out = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=3)
t = out["u"][:, :12]
names = ["backward: loop, K staging", "backward: spline + velocity box", "backward: row norms, stage constants (+ on entry: the batches' queue)",
         "lane-level certificates (upper + lower LP)", "batches: rebuild rows across lanes", "backward: take the batches' results (when entered), warm-start update",
         "batches: pick queue entries", "forward: prefetch, update, output staging", "forward: spline evaluation",
         "forward: 1-variable LP (15 divisions)", "batches: simplex walk (predict_upper_lp)", "batches: full Seidel iteration"]
m = t.mean(0)
print("cycles per wave (mean over lanes) for B=%d d=%d N=%d:" % (B, d, N))
for n, v in sorted(zip(names, m), key=lambda p: -p[1]):
    print("  %-72s %12.0f  %5.1f%%   per stage %8.0f" % (n, v, 100 * v / m.sum(), v / N))
print("  total %.0f cycles = %.3f ms at 2.4 GHz" % (m.sum(), m.sum() / 2.4e6))
it = out["u"][:, 12:14]
print("  forward LP: the wave repeats it while any lane retries (reference: lower x_i and solve again): %.4f passes per stage; "
      "lanes with a retry anywhere: %.2f%%, retries per trajectory %.3f" % (it[:, 0].mean() / N, 100 * (it[:, 1] > 0).mean(), it[:, 1].mean()))
be = out["u"][:, 14:16]
bt = m[[4, 6, 10, 11]].sum()  # (rows, queue picks, walk, iteration: the parts that only run on entry)
print("  batches: entered at %.1f%% of the stages (%.2f LPs per entry); %.0f cycles per entry" % (
    100 * be[:, 0].mean() / N, be[:, 1].mean() / max(be[:, 0].mean(), 1e-9), bt / max(be[:, 0].mean(), 1e-9)))

#!/usr/bin/env python
"""Where kernel family 4 (one trajectory per wave) spends its cycles, and how often its certificates hold.

    python -m toppra_amd.build -DTPR_WAVE_TIMING -DTPR_CERT_DEV --out=build_dbg/libtoppra_wtim.so    (build container)
    python -m toppra_amd.build -DTPR_WAVE_STATS -DTPR_CERT_DEV --out=build_dbg/libtoppra_wstat.so
    TOPPRA_HIP_LIB=build_dbg/libtoppra_wtim.so python tools/gpu_wave_phases.py timing [B d N]         (GPU box)
    TOPPRA_HIP_LIB=build_dbg/libtoppra_wstat.so python tools/gpu_wave_phases.py stats [B d N]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toppra_amd import batch
mode = sys.argv[1]
B, d, N = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (1, 7, 100)
data = batch.make_synthetic_batch(B, d, N)
out = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=4)
out = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=4)
if mode == "timing":
    names = ["backward: build rows", "backward: warm pair certificate (upper)", "backward: proposal + certificate", "backward: full iteration (upper)",
             "backward: warm row certificate (lower)", "backward: search + certificate (lower)", "backward: full iteration (lower)",
             "backward: loop, K to LDS", "forward: build rows", "forward: 1-variable LP", "forward: update, staging", "prologue", "epilogue"]
    m = out["u"][:, :13].mean(0)
    print("cycles per wave for B=%d d=%d N=%d:" % (B, d, N))
    for n, v in zip(names, m):
        print("  %-45s %10.0f  %5.1f%%   per stage %8.0f" % (n, v, 100 * v / m.sum(), v / N))
    print("  total %.0f cycles = %.4f ms at 2.4 GHz" % (m.sum(), m.sum() / 2.4e6))
else:
    st = out["status"].astype(np.int64)
    print("B=%d d=%d N=%d per trajectory: full iterations upper %.3f lower %.3f, proposals %.3f (of %d stages)" % (
        B, d, N, (st & 0xfff).mean(), ((st >> 12) & 0xfff).mean(), ((st >> 24) & 0xff).mean(), N))

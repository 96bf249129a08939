#!/usr/bin/env python
"""Where the kernel families cross over in batch size (kernel ms at d dof, N = 200): family 4 (one trajectory per wave),
family 2 (rows across lanes), family 3 (one trajectory per lane).  tpr_kernels.hip's pick_variant is set from this table.

  python tools/gpu_crossover.py [d ...]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toppra_amd import batch

dofs = [int(a) for a in sys.argv[1:]] or [7]
N = 200
for d in dofs:
    for B in (2048, 4096, 6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576, 32768):
        data = batch.make_synthetic_batch(B, d, N)
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in data.items() if isinstance(v, np.ndarray)}
        out = {"sd2": torch.empty((B, N + 1), dtype=torch.float64, device="cuda"), "u": torch.empty((B, N), dtype=torch.float64, device="cuda"),
               "K": torch.empty((B, N + 1, 2), dtype=torch.float64, device="cuda"), "status": torch.empty(B, dtype=torch.int32, device="cuda")}
        ms = {}
        for v in (4, 2, 3, 0):
            if v == 3 and d > 13:
                continue
            if v == 4 and B > 16384:
                continue
            try:
                ms[v] = batch.solve_batch_timed(dev["coef"], dev["breaks"], dev["grid"], dev["vlim"], dev["alim"], out, 5, variant=v)
            except Exception as e:  # noqa: BLE001
                ms[v] = float("nan")
        best = min((m, v) for v, m in ms.items() if v != 0 and m == m)
        print("d %2d B %6d: " % (d, B) + "  ".join("v%d %.3f" % (v, m) for v, m in ms.items()) + "   best v%d%s" % (best[1], "" if abs(ms[0] - best[0]) < 0.05 * best[0] else "  <-- auto picks slower"), flush=True)

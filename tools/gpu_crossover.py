#!/usr/bin/env python
"""Kernel time of families 2 and 3 against the batch size (where should the default switch?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toppra_amd import batch
d, N = 7, 200
for B in (4096, 8192, 12288, 16384, 20480, 24576, 32768):
    data = batch.make_synthetic_batch(B, d, N)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in data.items() if isinstance(v, np.ndarray)}
    out = {"sd2": torch.empty((B, N + 1), dtype=torch.float64, device="cuda"), "u": torch.empty((B, N), dtype=torch.float64, device="cuda"),
           "K": torch.empty((B, N + 1, 2), dtype=torch.float64, device="cuda"), "status": torch.empty(B, dtype=torch.int32, device="cuda")}
    ms = {}
    for variant in (2, 3, 0):
        for _ in range(2):
            ms[variant] = batch.solve_batch_timed(dev["coef"], dev["breaks"], dev["grid"], dev["vlim"], dev["alim"], out, 5, variant=variant)
    print("B=%6d  family 2 %.3f ms   family 3 %.3f ms   default %.3f ms" % (B, ms[2], ms[3], ms[0]))

#!/usr/bin/env python
"""Kernel time of the headline launch for the library in TOPPRA_HIP_LIB, without any check of the results
(for timing experiments whose results are deliberately wrong, e.g. -DTPR_EXPERIMENT_NO_BATCH)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toppra_amd import batch
B, d, N = 65536, int(os.environ.get("TPR_TIME_DOF", "7")), 200
data = batch.make_synthetic_batch(B, d, N)
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in data.items() if isinstance(v, np.ndarray)}
out = {"sd2": torch.empty((B, N + 1), dtype=torch.float64, device="cuda"), "u": torch.empty((B, N), dtype=torch.float64, device="cuda"),
       "K": torch.empty((B, N + 1, 2), dtype=torch.float64, device="cuda"), "status": torch.empty(B, dtype=torch.int32, device="cuda")}
for _ in range(2):
    ms = batch.solve_batch_timed(dev["coef"], dev["breaks"], dev["grid"], dev["vlim"], dev["alim"], out, 5)
print(os.environ.get("TOPPRA_HIP_LIB", "default"), "dof", d, "kernel_ms %.3f" % ms)

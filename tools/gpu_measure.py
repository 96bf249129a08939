#!/usr/bin/env python
"""Secondary measurements quoted in DESIGN.md (not the bench line): the other BASELINE configs,
the PCIe-inclusive rate of the host-buffer entry, and the lane-per-trajectory kernel."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402


def device_rate(B, d, N, variant=0, reps=5):
    data = tb.make_synthetic_batch(B, d, N)
    dev = torch.device("cuda", 0)
    dv = {k: torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")}
    out = tb.solve_batch(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], variant=variant)
    torch.cuda.synchronize()
    ms = tb.solve_batch_timed(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], out, reps=reps, variant=variant)
    return {"B": B, "d": d, "N": N, "variant": variant, "kernel_ms": ms, "traj_per_s": B / ms * 1e3,
            "lps_per_s": 3 * N * B / ms * 1e3, "ok": float((out["status"] == 0).double().mean())}


def host_rate(B, d, N, reps=3):
    data = tb.make_synthetic_batch(B, d, N)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    tb.solve_batch(*args)
    t0 = time.perf_counter()
    for _ in range(reps):
        tb.solve_batch(*args)
    dt = (time.perf_counter() - t0) / reps
    return {"B": B, "d": d, "N": N, "host_ms": dt * 1e3, "traj_per_s_pcie_inclusive": B / dt}


def other_entries(B=65536, d=7, N=200):
    """TOPPRAsd, stand-alone controllable / feasible sets, robust config 4 (device tensors in and out)."""
    data = tb.make_synthetic_batch(B, d, N)
    dev = torch.device("cuda", 0)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    zero = torch.zeros(B, dtype=torch.float64, device=dev)

    def timed(fn, reps=10):  # wall time per call over 10 calls after two warm-up calls (the first launches of a process run 5-10 % slow)
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {
        "toppra_sd_65536x7x200_ms": timed(lambda: tb.solve_desired_duration_batch(*dv, 3.0)),
        "controllable_sets_65536x7x200_ms": timed(lambda: tb.controllable_sets_batch(*dv, zero, zero)),
        "feasible_sets_65536x7x200_ms": timed(lambda: tb.feasible_sets_batch(*dv)),
    }
    data4 = tb.make_synthetic_batch(16384, 7, 100)
    dv4 = [torch.from_numpy(np.ascontiguousarray(data4[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    res["robust_config4_16384x7x100_ms"] = timed(lambda: tb.robust_solve_batch(*dv4, [1e-3, 5e-2, 9e-3]))
    return res


if __name__ == "__main__":
    res = {
        "headline_65536x7x200": device_rate(65536, 7, 200),
        "c2_4096x7x200": device_rate(4096, 7, 200),
        "c2_4096x7x200_family3": device_rate(4096, 7, 200, variant=3),
        "b32768x7x200_family2": device_rate(32768, 7, 200, variant=2),
        "b32768x7x200_family3": device_rate(32768, 7, 200, variant=3),
        "headline_family2": device_rate(65536, 7, 200, variant=2),
        "c3_65536x6x500": device_rate(65536, 6, 500, reps=2),
        "headline_lane_kernel": device_rate(65536, 7, 200, variant=1, reps=2),
        "host_buffers_65536x7x200": host_rate(65536, 7, 200),
        "b262144x7x200": device_rate(262144, 7, 200, reps=2),
        "other_entries": other_entries(),
    }
    print(json.dumps(res, indent=1))

#!/usr/bin/env python
"""Every kernel that is not the headline one, launched a few times on device-resident inputs so that rocprofv3 can
attribute time and counters to it ("rocprof summaries for any kernel but the headline one"):

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sec -o run --output-format csv -- python tools/gpu_secondary_kernels.py
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d ... -- python ...

Without the profiler it prints the wall time per call (device tensors in and out, synchronised)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

dev = torch.device("cuda", 0)
REPS = int(os.environ.get("SEC_REPS", "3"))
ONLY = os.environ.get("SEC_ONLY", "")


def dev_args(data):
    return [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


def main():
    B, d, N = 65536, 7, 200
    data = tb.make_synthetic_batch(B, d, N)
    dv = dev_args(data)
    zero = torch.zeros(B, dtype=torch.float64, device=dev)
    res = {}
    cases = {
        "family2_default_65536x7x200": lambda: tb.solve_batch(*dv, variant=2),
        "family2_strict_65536x7x200": lambda: tb.solve_batch(*dv, variant=2, strict=True),
        "family4_4096x7x200": None,
        "feasible_sets_65536x7x200": lambda: tb.feasible_sets_batch(*dv),
        "feasible_sets_family2_65536x7x200": lambda: tb.feasible_sets_batch(*dv, variant=2),
        "controllable_sets_65536x7x200": lambda: tb.controllable_sets_batch(*dv, zero, zero),
        "toppra_sd_65536x7x200": lambda: tb.solve_desired_duration_batch(*dv, 3.0),
        "toppra_sd_family2_65536x7x200": lambda: tb.solve_desired_duration_batch(*dv, 3.0, variant=2),
    }
    d2 = tb.make_synthetic_batch(4096, 7, 200)
    dv2 = dev_args(d2)
    cases["family4_4096x7x200"] = lambda: tb.solve_batch(*dv2, variant=4)
    cases["family5_4096x7x200"] = lambda: tb.solve_batch(*dv2, variant=5)
    d1 = tb.make_synthetic_batch(1, 7, 100)
    dv1 = dev_args(d1)
    cases["family4_1x7x100"] = lambda: tb.solve_batch(*dv1, variant=4)
    d4 = tb.make_synthetic_batch(16384, 7, 100)
    dv4 = dev_args(d4)
    cases["robust_config4_16384x7x100"] = lambda: tb.robust_solve_batch(*dv4, [1e-3, 5e-2, 9e-3])
    d12 = tb.make_synthetic_batch(65536, 12, 200)
    dv12 = dev_args(d12)
    cases["family2_default_65536x12x200"] = lambda: tb.solve_batch(*dv12, variant=2)
    cases["family3_65536x12x200"] = lambda: tb.solve_batch(*dv12)
    cases["robust_collocation_16384x7x100"] = lambda: tb.robust_solve_batch(*dv4, [1e-3, 5e-2, 9e-3], interpolation=False)
    cases["robust_with_feasible_sets_16384x7x100"] = lambda: tb.robust_solve_batch(*dv4, [1e-3, 5e-2, 9e-3], want_X=True)
    # dense rows (any canonical-linear constraint list): the standard problem's own rows fed back as arrays
    cases["constraint_params_65536x7x200"] = lambda: tb.constraint_params_batch(*dv)
    rows = tb.constraint_params_batch(*dv)
    dense = (rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], dv[2][1:] - dv[2][:-1])
    cases["dense_rows_solve_65536x7x200"] = lambda: tb.solve_dense_batch(*dense)
    # f1 / f2: spline fit of the waypoints, ParametrizeSpline of the result, evaluation
    rng = np.random.default_rng(1)
    way = torch.from_numpy(rng.standard_normal((B, 5, d))).to(dev)
    knots = torch.linspace(0, 1, 5, dtype=torch.float64, device=dev)
    cases["spline_fit_65536x7x5pts"] = lambda: tb.spline_fit_batch(knots, way)
    sol = tb.solve_batch(*dv, want_sd=True, want_K=False, want_u=False)
    cases["param_spline_65536x7x200"] = lambda: tb.param_spline_batch(dv[0], dv[1], dv[2], sol["sd"])
    cases["param_spline_lapack_order_65536x7x200"] = lambda: tb.param_spline_batch(dv[0], dv[1], dv[2], sol["sd"], variant=2)
    sp = tb.param_spline_batch(dv[0], dv[1], dv[2], sol["sd"])
    times = torch.rand(B, 64, dtype=torch.float64, device=dev) * 2.0
    cases["ppoly_eval_65536x64"] = lambda: tb.ppoly_eval_batch(sp["coef"], sp["knot_times"], times, 0, sp["counts"])
    frac = torch.linspace(0, 1, 64, dtype=torch.float64, device=dev)
    cases["param_spline_sample_65536x64"] = lambda: tb.param_spline_sample_batch(dv[0], dv[1], dv[2], sol["sd"], frac)
    ts_us = tb.const_accel_times_batch(dv[2], sol["sd"])
    cases["const_accel_times_65536x200"] = lambda: tb.const_accel_times_batch(dv[2], sol["sd"])
    cases["const_accel_eval_65536x64"] = lambda: tb.const_accel_eval_batch(dv[0], dv[1], dv[2], sol["sd"], ts_us[0], ts_us[1], times, 0)
    for name, fn in cases.items():
        if ONLY and ONLY not in name:
            continue
        try:
            res[name + "_ms"] = timed(fn)
        except Exception as exc:  # noqa: BLE001
            res[name + "_ms"] = repr(exc)[:120]
        print("%-42s %s" % (name, res[name + "_ms"]), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

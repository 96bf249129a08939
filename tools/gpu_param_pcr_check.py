#!/usr/bin/env python
"""ParametrizeSpline: the knot-parallel kernel (variant 3, cyclic reduction) against the LAPACK-order kernels (variant 2
fused, variant 1 generic): knot times and counts bit for bit, the splines compared where they are used -- q, dq/dt,
d2q/dt2 evaluated inside every segment -- and the timings at the headline shape.
  python tools/gpu_param_pcr_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

dev = torch.device("cuda", 0)


def harsh_profiles(rng, B, N):
    sd = 0.2 + 3 * rng.random((B, N + 1))
    hole = rng.random((B, N + 1)) < 0.08
    sd[hole] = 1e12
    sd[:, 1:][hole[:, :-1] & (rng.random((B, N)) < 0.7)] = 1e12
    still = rng.random((B, N + 1)) < 0.05
    sd[still] = 0.0
    sd[:, 1:][still[:, :-1] & (rng.random((B, N)) < 0.5)] = 0.0
    sd[0] = 1e12
    if B > 3:
        sd[1, 1:] = 1e12
        sd[1, -1] = 1.0
        sd[2, N // 2:] = np.nan
        sd[3, 0] = 0.0
    return sd


def deviation(a, b):
    """max over segments in use of |q_a - q_b| (orders 0, 1, 2) at 0, 1/2, 1 of the segment, absolute and relative to the
    largest |value| of that order on the trajectory"""
    cnt = a["counts"].astype(int)
    kt = a["knot_times"]
    ca, cb = a["coef"], b["coef"]
    N = ca.shape[2]
    seg = np.arange(N)[None, :] <= (cnt[:, None] - 2)
    dx = np.where(seg, kt[:, 1:] - kt[:, :-1], 0.0)
    out = []
    for order in (0, 1, 2):
        diffs, scale = [], 1e-300
        for f in (0.0, 0.5, 1.0):
            x = (f * dx)[:, :, None]
            vals = []
            for c in (ca, cb):
                if order == 0:
                    val = ((c[:, 3] + c[:, 2] * x) + c[:, 1] * x * x) + c[:, 0] * x * x * x
                elif order == 1:
                    val = (c[:, 2] + 2 * c[:, 1] * x) + 3 * c[:, 0] * x * x
                else:
                    val = 2 * c[:, 1] + 6 * c[:, 0] * x
                vals.append(np.where(seg[:, :, None], val, 0.0))
            fin = np.isfinite(vals[1])
            assert np.array_equal(np.isfinite(vals[0]), fin)  # the same trajectories / segments are NaN
            diffs.append(np.where(fin, np.abs(vals[0] - vals[1]), 0.0))
            scale = np.maximum(scale, np.where(fin, np.abs(vals[1]), 0.0).max(axis=(1, 2), keepdims=True))
        out.append((max(float(df.max()) for df in diffs), max(float((df / scale).max()) for df in diffs)))
    return out


def main():
    bad = 0
    for B, d, N in ((4096, 7, 200), (300, 7, 200), (70, 1, 64), (64, 8, 255), (33, 2, 256), (40, 6, 500), (20, 3, 1000), (100, 5, 3), (50, 5, 1), (17, 4, 2), (40, 12, 150), (24, 16, 100)):
        rng = np.random.default_rng(1000 * d + N)
        data = tb.make_synthetic_batch(B, d, N, seed=5 + d)
        for kind in ("solved", "harsh"):
            if kind == "solved":
                sd = tb.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], want_sd=True)["sd"]
            else:
                sd = harsh_profiles(rng, B, N)
            new = tb.param_spline_batch(data["coef"], data["breaks"], data["grid"], sd, variant=3)
            old = tb.param_spline_batch(data["coef"], data["breaks"], data["grid"], sd, variant=2)
            same = all(np.array_equal(new[k], old[k], equal_nan=True) for k in ("counts", "knot_times"))
            dv = deviation(new, old)
            cdiff = np.nanmax(np.abs(new["coef"] - old["coef"]) / np.maximum(np.abs(old["coef"]), 1e-300)) if new["coef"].size else 0.0
            print("B %4d d %d N %4d %-6s times/counts identical %s; |dq| abs/rel %.1e/%.1e  |dq'| %.1e/%.1e  |dq''| %.1e/%.1e; coef rel %.1e"
                  % (B, d, N, kind, same, dv[0][0], dv[0][1], dv[1][0], dv[1][1], dv[2][0], dv[2][1], cdiff), flush=True)
            bad += not same
    B, d, N = 65536, 7, 200
    data = tb.make_synthetic_batch(B, d, N)
    dvt = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    sol = tb.solve_batch(*dvt, want_sd=True, want_K=False, want_u=False)
    for shape_note, args in (("65536 x 7 x 200", (dvt[0], dvt[1], dvt[2], sol["sd"])),):
        for variant in (3, 2, 0):
            tb.param_spline_batch(*args, variant=variant)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                tb.param_spline_batch(*args, variant=variant)
            e1.record()
            torch.cuda.synchronize()
            print("time %s variant %d: %.3f ms per call" % (shape_note, variant, e0.elapsed_time(e1) / 5), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

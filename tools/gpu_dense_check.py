#!/usr/bin/env python
"""The dense-row entries (tpr_*_dense_batch: any canonical-linear constraint list, the reference's full Seidel iteration on
rows given as arrays) against the fused kernels: the rows tpr_constraint_params_batch produces for the standard
velocity + acceleration problem, fed back as dense arrays, must give the same bits -- parameterization, controllable sets,
feasible sets; Interpolation, Collocation, acceleration only, boundary velocities, 1..16 dof -- and the timing at the
headline shape (144 KB of rows per trajectory: the HBM-heavy form of the path).
  python tools/gpu_dense_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from toppra_amd import batch as tb  # noqa: E402

bad = checks = 0


def same(label, got, want):
    global bad, checks
    checks += 1
    got, want = np.asarray(got), np.asarray(want)
    eq = (got == want) | (np.isnan(got) & np.isnan(want))
    if not eq.all():
        bad += 1
        rows = np.flatnonzero(~eq.reshape(len(got), -1).all(axis=1))
        print("MISMATCH %-60s %d trajectories differ (first %d)" % (label, len(rows), rows[0]), flush=True)
    else:
        print("ok       %-60s" % label, flush=True)


def main():
    rng = np.random.default_rng(7)
    for B, d, N, interp, vel in ((512, 7, 200, True, True), (300, 7, 60, False, True), (256, 3, 50, True, False), (100, 1, 30, True, True),
                                 (64, 8, 40, True, True), (64, 9, 40, True, True), (48, 12, 30, False, True), (32, 16, 25, True, True),
                                 (200, 5, 3, True, True), (7, 4, 1, True, True)):
        data = tb.make_synthetic_batch(B, d, N, seed=100 + d + N)
        vlim = data["vlim"] if vel else None
        sd0 = 0.3 * rng.random(B) * (rng.random(B) < 0.5)
        sd1 = 0.3 * rng.random(B) * (rng.random(B) < 0.5)
        args = (data["coef"], data["breaks"], data["grid"], vlim, data["alim"])
        rows = tb.constraint_params_batch(*args, interp)
        deltas = np.diff(data["grid"])
        dense = (rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], deltas)
        tag = "B%d d%d N%d %s%s" % (B, d, N, "interp" if interp else "colloc", "" if vel else " acc only")
        ref = tb.solve_batch(*args, sd0, sd1, interp, want_sd=True)
        got = tb.solve_dense_batch(*dense, sd0, sd1, want_sd=True)
        for k in ("K", "sd2", "sd", "u", "status"):
            same("%s parameterization %s" % (tag, k), got[k], ref[k])
        same("%s controllable sets" % tag, tb.controllable_sets_dense_batch(*dense, 0.1 * sd1, sd1 + 0.2),
             tb.controllable_sets_batch(*args, 0.1 * sd1, sd1 + 0.2, interp))
        same("%s feasible sets" % tag, tb.feasible_sets_dense_batch(*dense), tb.feasible_sets_batch(*args, interp))
    # timing at the headline shape, device-resident rows
    dev = torch.device("cuda", 0)
    B, d, N = 65536, 7, 200
    data = tb.make_synthetic_batch(B, d, N)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    rows = tb.constraint_params_batch(*dv)
    deltas = torch.from_numpy(np.diff(data["grid"])).to(dev)
    dense = (rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], deltas)
    ref = tb.solve_batch(*dv)
    got = tb.solve_dense_batch(*dense)
    same("65536 x 7 x 200 parameterization sd2 (device)", got["sd2"].cpu().numpy(), ref["sd2"].cpu().numpy())
    nbytes = 8 * B * ((N + 1) * (3 * 30 + 4) + N) * 2  # rows + boxes + deltas, read by the backward and by the forward scan
    for name, fn in (("dense parameterization", lambda: tb.solve_dense_batch(*dense)),
                     ("dense feasible sets", lambda: tb.feasible_sets_dense_batch(*dense)),
                     ("fused strict (family 2, full iteration)", lambda: tb.solve_batch(*dv, variant=2, strict=True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("time     %-42s %.2f ms per call%s" % (name, ms, "  (rows read: %.1f GB -> %.0f GB/s)" % (nbytes / 1e9, nbytes / ms / 1e6)
                                                       if name == "dense parameterization" else ""), flush=True)
    print("checks %d, mismatching %d" % (checks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

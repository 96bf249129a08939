"""Solve a batch of random 7-DoF retiming problems in one launch and sample the trajectories --
waypoints -> GPU spline fit -> TOPP-RA -> constant-acceleration parametrization, all on the MI355X.

    python examples/batch_kinematics.py [--batch 4096] [--desired-duration 3.0]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toppra_amd as ta  # noqa: E402
from toppra_amd import batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--desired-duration", type=float, default=None, help="use TOPPRAsd with this duration [s]")
    args = ap.parse_args()
    B, d, N = args.batch, 7, 200
    rng = np.random.default_rng(0)
    knots = np.linspace(0, 1, 5)
    way = rng.standard_normal((B, 5, d))
    vmax, amax = 10 + 20 * rng.random((B, d)), 10 + 2 * rng.random((B, d))
    vlim, alim = np.stack([-vmax, vmax], -1), np.stack([-amax, amax], -1)
    grid = np.linspace(0, 1, N + 1)

    t0 = time.perf_counter()
    bt = ta.algorithm.BatchTOPPRA.from_waypoints(knots, way, grid, vlim, alim)
    if args.desired_duration is None:
        out = bt.compute_parameterization()
    else:
        out = bt.compute_parameterization_sd(args.desired_duration)
    ts, us = batch.const_accel_times_batch(grid, out["sd"])
    samples = np.linspace(0, 1, 50)[None, :] * ts[:, -1:]
    q = batch.const_accel_eval_batch(bt.coef, bt.breaks, grid, out["sd"], ts, us, samples, order=0)
    dt = time.perf_counter() - t0
    codes = ta.algorithm.BatchTOPPRA.return_codes(out["status"])
    print("%d trajectories in %.1f ms (host buffers, incl. transfers); %d Ok" % (B, dt * 1e3, sum(c.name == "Ok" for c in codes)))
    print("durations: min %.3f  mean %.3f  max %.3f s" % (ts[:, -1].min(), ts[:, -1].mean(), ts[:, -1].max()))
    print("q(t) of trajectory 0 at 5 instants:\n", np.round(q[0, ::12], 4))
    if args.desired_duration is None:
        # the reference's default output: a cubic spline in time through the gridpoints (ParametrizeSpline), GPU end to end
        traj = bt.compute_trajectory()
        dur = traj.duration
        qs = traj(np.linspace(0, 1, 50)[None, :] * dur[:, None], order=1)
        print("ParametrizeSpline: durations mean %.3f s, max |dq/dt| of trajectory 0 = %.3f" % (dur.mean(), np.abs(qs[0]).max()))


if __name__ == "__main__":
    main()

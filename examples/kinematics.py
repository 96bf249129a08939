"""Retime a 7-DoF spline path subject to joint velocity and acceleration limits -- the scenario of
the reference's examples/plot_kinematics.py (BASELINE config 1), on the MI355X.

    python examples/kinematics.py [--robust]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toppra_amd as ta  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robust", action="store_true", help="robust acceleration constraint (plot_robust_kinematics.py)")
    ap.add_argument("-N", type=int, default=None, help="number of grid segments (default: automatic)")
    args = ap.parse_args()

    np.random.seed(9)
    way_pts = np.random.randn(5, 7)
    vlim_ = 10 + np.random.rand(7) * 20
    alim_ = 10 + np.random.rand(7) * 2
    path = ta.SplineInterpolator(np.linspace(0, 1, 5), way_pts)
    pc_vel = ta.constraint.JointVelocityConstraint(np.vstack((-vlim_, vlim_)).T)
    pc_acc = ta.constraint.JointAccelerationConstraint(np.vstack((-alim_, alim_)).T)
    if args.robust:
        pc_acc = ta.constraint.RobustLinearConstraint(pc_acc, [1e-3, 5e-2, 9e-3], 1)
    grid = None if args.N is None else np.linspace(0, 1, args.N + 1)
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=grid)
    traj = inst.compute_trajectory(0, 0)
    data = inst.problem_data
    print("return code :", data.return_code)
    print("gridpoints  :", len(data.gridpoints))
    print("duration    : %.4f s" % traj.duration)
    ts = np.linspace(0, traj.duration, 5)
    print("q(t) samples:\n", np.round(traj(ts), 4))
    print("max |qd|/vlim: %.4f   max |qdd|/alim: %.4f" % (
        np.max(np.abs(traj(np.linspace(0, traj.duration, 400), 1)) / vlim_),
        np.max(np.abs(traj(np.linspace(0, traj.duration, 400), 2)) / alim_)))


if __name__ == "__main__":
    main()

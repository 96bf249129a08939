"""toppra_amd -- MI355X-native batched TOPP-RA (the reference's seidel hot path on HIP).

Public surface mirrors the reference package for this path::

    import toppra_amd as ta
    path = ta.SplineInterpolator(ss, waypoints)
    pc_vel = ta.constraint.JointVelocityConstraint(vlim)
    pc_acc = ta.constraint.JointAccelerationConstraint(alim)
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=grid)
    sdd, sd, _ = inst.compute_parameterization(0, 0)

plus the batched entry points (``ta.algorithm.BatchTOPPRA``, ``ta.batch.solve_batch``) that solve B
trajectories per kernel launch.  All arithmetic runs in ``libtoppra_hip.so`` (hand-written HIP for
gfx950 behind the C-ABI of ``include/toppra_hip.h``); there is no CPU fallback.
"""
import logging

from . import algorithm, batch, constants, constraint, exceptions, interpolator, parametrizer, solverwrapper
from .interpolator import SplineInterpolator
from .parametrizer import ParametrizeConstAccel, ParametrizeSpline

logging.getLogger("toppra_amd").addHandler(logging.NullHandler())

__all__ = ["algorithm", "batch", "constants", "constraint", "exceptions", "interpolator", "parametrizer",
           "solverwrapper", "SplineInterpolator", "ParametrizeConstAccel", "ParametrizeSpline"]

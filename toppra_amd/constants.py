"""Numerical constants that are part of the path's observable behaviour.

Two families exist in the reference and both matter for parity (trap 4):
the Python-level ones (toppra/constants.py:15-47) used by the scans, and the solver-level ones
compiled into the Cython seidel module (cy_seidel_solverwrapper.pyx:17-27), which live in
csrc/tpr_device.hpp on the device side.
"""
SUPERTINY = 1e-10
TINY = 1e-8
SMALL = 1e-5
LARGE = 1000.0
VERYLARGE = 1e8
INFTY = 1e16

MAX_TRIES = 10          # forward-pass retries (reachability_algorithm.py:315-343)
JVEL_MAXSD = 1e8        # cap on sd in the velocity constraint (a C float in the reference)
CVXPY_MAXX = 10000      # x / x_next box of compute_feasible_sets
CVXPY_MAXU = 10000

# solver-level (seidel) constants, mirrored for documentation; the kernels use their own copy
SEIDEL_TINY = 1e-10
SEIDEL_SMALL = 1e-8
SEIDEL_VAR_MIN = -1e8
SEIDEL_VAR_MAX = 1e8
SEIDEL_INF = 1e10

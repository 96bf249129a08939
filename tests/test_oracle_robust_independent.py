"""Config 4 (robust TOPP-RA), row f4: the CPU restatement of the product's method
(oracle/seidel_oracle.c ``orc_robust_*``: closed-form u-interval per row + bisection on x) against an
INDEPENDENT solution of the same second-order-cone stage problems (oracle/robust_independent.py:
cutting planes + exhaustive vertex enumeration, problems rebuilt from
ecos_solverwrapper.py:94-188 / conic_constraint.py:19-26,95-124 with scipy/numpy only).

Tolerance: the stated bar is 1e-7 on K and X (ECOS itself stops at ~1e-8); the
two methods actually agree to ~1e-12, asserted here at 1e-9.  The -m gpu twin of this test
(tests/test_gpu_robust.py::test_kernel_matches_independent_solver) holds the HIP kernels to the same
checker."""
import numpy as np
import pytest

from oracle import robust_independent as ri
from toppra_amd import batch

ELL = [1e-3, 5e-2, 9e-3]  # examples/plot_robust_kinematics.py:26-28

CASES = [
    # B, d, N, interpolation, ellipsoid, non-zero sd_end, velocity constraint
    (12, 7, 100, True, ELL, False, True),
    (10, 3, 40, False, ELL, True, True),       # Collocation, non-zero sd_end
    (8, 6, 150, True, [1e-2, 1e-1, 5e-2], True, True),
    (8, 4, 60, True, ELL, False, False),       # no velocity constraint: the +-ECOS_INFTY stand-ins bind
    (6, 2, 50, True, [0.0, 0.0, 0.0], False, True),  # zero ellipsoid: the plain LP
]


@pytest.mark.parametrize("B,d,N,interp,ell,end_vel,has_vel", CASES)
def test_oracle_robust_matches_independent_solver(oracle, B, d, N, interp, ell, end_vel, has_vel):
    data = batch.make_synthetic_batch(B, d, N, seed=100 + d)
    rng = np.random.default_rng(d)
    sd_end = 0.3 * rng.random(B) if end_vel else None
    vlim = data["vlim"] if has_vel else None
    flags = (oracle.FLAG_VEL if has_vel else 0) | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
    out = oracle.robust_solve_batch(data["coef"], data["breaks"], data["grid"], vlim, data["alim"], ell,
                                    None, sd_end, flags=flags)
    assert (out["status"] == 0).mean() >= 0.7
    d2 = dict(data, vlim=vlim)
    agg = ri.check_batch(d2, ell, out, interp, stride=3, tol_x=1e-9)
    assert agg["stages"] >= N // 3 * int((out["status"] == 0).sum()) - B
    assert agg["K"] <= 1e-9 and agg["X"] <= 1e-9


def test_uncontrollable_trajectories_are_infeasible_for_the_independent_solver_too(oracle):
    data = batch.make_synthetic_batch(16, 5, 60, seed=77)
    sd_end = np.full(16, 40.0)  # far above what the limits allow at the last gridpoint
    flags = oracle.FLAG_VEL | oracle.FLAG_ACC | oracle.FLAG_INTERP
    out = oracle.robust_solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], ELL,
                                    None, sd_end, flags=flags)
    assert (out["status"] == 1).all()
    agg = ri.check_batch(data, ELL, out, True)
    assert agg["failed_confirmed"] >= 12


def test_stage_solver_known_answers():
    """The checker itself on problems with answers known in closed form."""
    # one cone row  u + ||(0, 0, rc)|| <= 0 and x free in a box: max x = box, u <= -rc
    z = ri.solve_stage([1.0], [0.0], [0.0], [0.0, 0.0, 0.5], [[0, 1.0], [0, -1.0]], [2.0, 1.0], [1e-9, 1.0])
    assert abs(z[1] - 2.0) < 1e-12 and abs(z[0] + 0.5) < 1e-9
    # disc-like row: -x + 0.1 + ||(u, 0, 0)|| <= 0  i.e. x >= 0.1 + |u|; min x = 0.1 at u = 0
    z = ri.solve_stage([0.0], [-1.0], [0.1], [1.0, 0.0, 0.0], [[0, 1.0], [0, -1.0]], [5.0, 5.0], [0.0, -1.0])
    assert abs(z[1] - 0.1) < 1e-10 and abs(z[0]) < 1e-6
    # infeasible: x <= -1 and x >= 0
    assert ri.solve_stage([0.0], [1.0], [1.0], [0.0, 0.0, 0.0], [[0, -1.0]], [0.0], [0.0, 1.0]) is None

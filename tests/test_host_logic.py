"""Host-side mirror of the reference interface: argument validation, error behaviour, gridpoint
proposal, input marshalling.  No GPU needed (nothing here launches a kernel)."""
import numpy as np
import pytest

import toppra_amd as ta
from toppra_amd import _capi, batch
from toppra_amd.algorithm import ParameterizationReturnCode
from tests.helpers import golden


@pytest.fixture
def example():
    fx = golden("example_kinematics_seed9")
    path = ta.SplineInterpolator(fx["knots"], fx["way_pts"])
    pc_vel = ta.constraint.JointVelocityConstraint(fx["vlim"][0])
    pc_acc = ta.constraint.JointAccelerationConstraint(fx["alim"][0])
    return fx, path, pc_vel, pc_acc


def test_spline_tables_match_reference_coefficients(example):
    fx, path, _, _ = example
    coef, breaks = ta.interpolator.spline_tables(path)
    assert np.array_equal(coef, fx["coef"][0]) and np.array_equal(breaks, fx["breaks"])
    assert path.dof == 7 and np.array_equal(path.path_interval, [0.0, 1.0])
    assert np.array_equal(path(fx["n100_grid"], 1), fx["n100_qs"])


def test_batched_fit_is_bitwise_per_trajectory_fit():
    rng = np.random.default_rng(0)
    way = rng.standard_normal((6, 5, 3))
    knots = np.linspace(0, 1, 5)
    coef, breaks = batch.spline_coefficients(knots, way)
    for b in range(6):
        assert np.array_equal(coef[b], ta.SplineInterpolator(knots, way[b]).cspl.c)


def test_propose_gridpoints_matches_reference(example):
    fx, path, _, _ = example
    grid = ta.interpolator.propose_gridpoints(path, max_err_threshold=1e-3, min_nb_points=100)
    assert np.array_equal(np.asarray(grid), fx["auto_grid"])


def test_constructor_validation(example):
    fx, path, pc_vel, pc_acc = example
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path)  # automatic gridpoints, lazy device init
    assert np.array_equal(inst.gridpoints, fx["auto_grid"])
    assert inst.solver_wrapper.get_no_stages() == len(fx["auto_grid"]) - 1
    assert inst.solver_wrapper.get_no_vars() == 2
    assert np.array_equal(inst.solver_wrapper.get_deltas(), np.diff(fx["auto_grid"]))
    assert inst.problem_data.return_code == ParameterizationReturnCode.ErrUnknown
    with pytest.raises(ValueError):
        ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=np.linspace(0, 0.9, 10))
    with pytest.raises(ValueError):
        ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=[0, 0.5, 0.4, 1.0])
    with pytest.raises(AssertionError):
        ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=np.linspace(0, 1, 11), solver_wrapper="qpoases")
    with pytest.raises(ta.exceptions.BadInputVelocities):
        ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=np.linspace(0, 1, 11)).compute_parameterization(-1, 0)


def test_constraint_validation():
    with pytest.raises(ValueError):
        ta.constraint.JointVelocityConstraint([[1.0, -1.0]])
    with pytest.raises(ValueError):
        ta.constraint.JointVelocityConstraint([np.nan, 1.0])
    c = ta.constraint.JointAccelerationConstraint([1.0, 2.0])
    assert c.alim.tolist() == [[-1, 1], [-2, 2]] and c.identical
    assert c.get_discretization_type() == ta.constraint.DiscretizationType.Interpolation
    c.set_discretization_type(0)
    assert c.get_discretization_type() == ta.constraint.DiscretizationType.Collocation
    path3 = ta.SplineInterpolator([0, 0.5, 1], np.zeros((3, 3)))
    with pytest.raises(ValueError):
        ta.solverwrapper.hipSeidelWrapper([c], path3, np.linspace(0, 1, 5))

    class Conic(ta.constraint.Constraint):
        constraint_type = ta.constraint.ConstraintType.CanonicalConic
        dof = 3
    with pytest.raises(NotImplementedError):
        ta.solverwrapper.hipSeidelWrapper([Conic()], path3, np.linspace(0, 1, 5))


def test_make_problem_flags_and_shapes():
    data = batch.make_synthetic_batch(5, 4, 20)
    p, keep = _capi.make_problem(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    assert (p.B, p.d, p.nseg, p.N) == (5, 4, 4, 20)
    assert p.flags == _capi.HAS_VELOCITY | _capi.HAS_ACCELERATION | _capi.ACC_INTERPOLATION
    grid2 = np.tile(data["grid"], (5, 1))
    p2, _ = _capi.make_problem(data["coef"], data["breaks"], grid2, None, data["alim"], interpolation=False)
    assert p2.flags == _capi.HAS_ACCELERATION | _capi.GRID_PER_TRAJ
    with pytest.raises(ValueError):
        _capi.make_problem(data["coef"], data["breaks"], data["grid"], data["vlim"][:, :3], data["alim"])
    with pytest.raises(ValueError):
        _capi.make_problem(data["coef"][:, :3], data["breaks"], data["grid"], None, data["alim"])


def test_parametrizers_on_golden_profile(example):
    fx, path, _, _ = example
    traj = ta.ParametrizeConstAccel(path, fx["n100_grid"], fx["n100_sd"])
    assert traj.path_interval[0] == 0 and traj.duration > 0
    np.testing.assert_allclose(traj(0.0), path(0.0))
    np.testing.assert_allclose(traj(traj.duration), path(1.0), atol=1e-9)
    np.testing.assert_allclose(traj(np.array([0.0, traj.duration]), 1), 0.0, atol=1e-9)
    spl = ta.ParametrizeSpline(path, fx["n100_grid"], fx["n100_sd"])
    assert abs(spl.duration - traj.duration) < 1e-9
    np.testing.assert_allclose(spl(spl.duration), path(1.0), atol=1e-9)


def test_parametrizers_match_reference_outputs(example):
    """Host parametrizers vs the reference's own ParametrizeConstAccel / ParametrizeSpline outputs
    (tests/golden, generated by tools/make_golden.py)."""
    fx, path, _, _ = example
    ca = ta.ParametrizeConstAccel(path, fx["n100_grid"], fx["n100_sd"])
    assert np.array_equal(ca._ts, fx["ca_ts"]) and np.array_equal(ca._us, fx["ca_us"])
    for order in (0, 1, 2):
        np.testing.assert_allclose(ca(fx["ca_times"], order), fx["ca_q%d" % order], rtol=1e-12, atol=1e-12)
    sp = ta.ParametrizeSpline(path, fx["n100_grid"], fx["n100_sd"])
    assert sp.duration == float(fx["spl_duration"])
    for order in (0, 1, 2):
        np.testing.assert_allclose(sp(fx["spl_times"], order), fx["spl_q%d" % order], rtol=1e-12, atol=1e-11)


def test_robust_constraint_validation():
    acc = ta.constraint.JointAccelerationConstraint([1.0, 2.0])
    with pytest.raises(ValueError):
        ta.constraint.RobustLinearConstraint(acc, [1e-3, -1.0, 0.0])
    rc = ta.constraint.RobustLinearConstraint(acc, [0.1, 0.2, 0.3], 1)
    assert rc.get_dof() == 2 and rc.get_discretization_type() == ta.constraint.DiscretizationType.Interpolation
    assert rc.get_constraint_type() == ta.constraint.ConstraintType.CanonicalConic

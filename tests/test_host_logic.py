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


# ---- argument validation at the ctypes boundary (raw pointers + sizes cross the C-ABI) ------
def _problem_arrays(B=4, d=3, N=10, nseg=2):
    rng = np.random.default_rng(0)
    return dict(coef=rng.standard_normal((B, 4, nseg, d)), breaks=np.linspace(0, 1, nseg + 1),
                grid=np.linspace(0, 1, N + 1), vlim=np.tile([[-1.0, 1.0]], (B, d, 1)), alim=np.tile([[-2.0, 2.0]], (B, d, 1)))


def test_make_problem_broadcasts_scalar_boundary_velocities():
    from toppra_amd import _capi
    a = _problem_arrays()
    p, keep = _capi.make_problem(a["coef"], a["breaks"], a["grid"], a["vlim"], a["alim"], 0.3, np.float64(0.1))
    sd_start, sd_end = keep[-2], keep[-1]
    assert sd_start.shape == (4,) and np.all(sd_start == 0.3) and sd_end.shape == (4,) and np.all(sd_end == 0.1)
    assert p.sd_start == sd_start.ctypes.data and p.B == 4 and p.N == 10


@pytest.mark.parametrize("bad", [
    dict(sd_start=np.zeros(3)),                       # wrong length: would be read out of bounds
    dict(sd_end=np.zeros((4, 1))),
    dict(grid=np.tile(np.linspace(0, 1, 11), (2, 1))),  # 2-D grid whose leading dim is not B
    dict(breaks=np.tile(np.linspace(0, 1, 3), (3, 1))),
    dict(grid=np.array([0.0, 0.5, 0.5, 1.0])),          # not strictly increasing
    dict(grid=np.array([0.0])),
    dict(vlim=np.zeros((4, 2, 2))),
    dict(alim=np.zeros((3, 3, 2))),
    dict(coef=np.zeros((4, 3, 2, 3))),
])
def test_make_problem_rejects_bad_shapes(bad):
    from toppra_amd import _capi
    a = _problem_arrays()
    sd = {k: bad.pop(k) for k in ("sd_start", "sd_end") if k in bad}
    a.update(bad)
    with pytest.raises(ValueError):
        _capi.make_problem(a["coef"], a["breaks"], a["grid"], a["vlim"], a["alim"], sd.get("sd_start"), sd.get("sd_end"))


def test_shard_problem_slices_only_per_trajectory_keys():
    from toppra_amd import distributed
    B = 5
    arrays = dict(coef=np.zeros((B, 4, 4, 2)), vlim=np.zeros((B, 2, 2)), alim=np.zeros((B, 2, 2)), knots=np.arange(float(B)),
                  grid=np.linspace(0, 1, B), breaks=np.linspace(0, 1, 5), sd_end=np.arange(float(B)))
    shard, (lo, hi) = distributed.shard_problem(arrays, 2, 1)
    assert (lo, hi) == (3, 5) and shard["coef"].shape[0] == 2 and np.array_equal(shard["sd_end"], [3.0, 4.0])
    # shared arrays whose length happens to equal B are NOT sliced
    assert shard["knots"].shape == (B,) and shard["grid"].shape == (B,)
    arrays["grid"] = np.tile(np.linspace(0, 1, 7), (B, 1))
    assert distributed.shard_problem(arrays, 2, 0)[0]["grid"].shape == (3, 7)
    arrays["vlim"] = np.zeros((B - 1, 2, 2))
    with pytest.raises(ValueError):
        distributed.shard_problem(arrays, 2, 0)

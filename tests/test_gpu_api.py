"""The drop-in Python surface on the GPU: reads like the reference's own tests
(tests/tests/retime/test_retime_basic.py, test_correct_velocity.py,
tests/tests/solverwrapper/test_basic_can_linear.py, tests/tests/constraint/*)."""
import numpy as np
import pytest

import toppra_amd as ta
from tests.helpers import assert_same, golden
from toppra_amd import batch
from toppra_amd.algorithm import ParameterizationReturnCode

pytestmark = pytest.mark.gpu


@pytest.fixture
def example():
    fx = golden("example_kinematics_seed9")
    path = ta.SplineInterpolator(fx["knots"], fx["way_pts"])
    pc_vel = ta.constraint.JointVelocityConstraint(fx["vlim"][0])
    pc_acc = ta.constraint.JointAccelerationConstraint(fx["alim"][0])
    return fx, path, pc_vel, pc_acc


def test_toppra_compute_parameterization(gpu, example):
    fx, path, pc_vel, pc_acc = example
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=fx["n100_grid"], solver_wrapper="hip")
    sdd, sd, v, K = inst.compute_parameterization(0, 0, return_data=True)
    assert inst.problem_data.return_code == ParameterizationReturnCode.Ok
    assert sd[0] == 0 and sd[-1] == 0 and v.shape == (100, 0)
    assert_same(sd, fx["n100_sd"], "sd_vec")
    assert_same(sdd, fx["n100_u"], "sdd_vec")
    assert_same(K, fx["n100_K"], "K")
    assert_same(inst.problem_data.K, fx["n100_K"], "problem_data.K")
    assert_same(inst.compute_feasible_sets(), fx["n100_X"], "X")
    assert_same(inst.compute_controllable_sets(0, 0), fx["n100_K"], "controllable")
    assert np.all(K >= 0) and not np.isnan(K).any()


def test_toppra_auto_gridpoints_and_trajectory(gpu, example):
    fx, path, pc_vel, pc_acc = example
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path)  # examples/plot_kinematics.py:38-48
    traj = inst.compute_trajectory()
    assert_same(inst.problem_data.sd_vec, fx["auto_sd"], "sd_vec")
    assert traj is not None and abs(traj.duration - 3.5248) < 1e-3  # SURVEY.md section 8(c) probe
    ts = np.linspace(0, traj.duration, 200)
    qd, qdd = traj(ts, 1), traj(ts, 2)
    assert np.all(np.abs(qd) <= fx["vlim"][0][:, 1] * 1.02)
    np.testing.assert_allclose(traj(0), path(0), atol=1e-9)
    np.testing.assert_allclose(traj(traj.duration), path(1), atol=1e-9)
    ca = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, parametrizer="ParametrizeConstAccel").compute_trajectory()
    assert abs(ca.duration - traj.duration) < 1e-6
    assert np.all(np.abs(ca(np.linspace(0, ca.duration, 300), 2)) <= fx["alim"][0][:, 1] * 1.05)


@pytest.mark.parametrize("sd_start,sd_end", [(0.1, 0.05), (0.0, 0.2)])
def test_boundary_velocities(gpu, example, sd_start, sd_end):
    fx, path, pc_vel, pc_acc = example
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=fx["n100_grid"])
    sdd, sd, _ = inst.compute_parameterization(sd_start, sd_end)
    np.testing.assert_allclose(sd[0], sd_start, atol=1e-7)
    np.testing.assert_allclose(sd[-1], sd_end, atol=1e-7)


def test_uncontrollable_start_returns_none(gpu, example):
    fx, path, pc_vel, pc_acc = example
    inst = ta.algorithm.TOPPRA([pc_vel, pc_acc], path, gridpoints=fx["n100_grid"])
    sdd, sd, v, K = inst.compute_parameterization(50.0, 0, return_data=True)
    assert sdd is None and sd is None and v is None
    assert inst.problem_data.return_code == ParameterizationReturnCode.FailUncontrollable
    assert_same(K, fx["n100_K"], "K")
    assert inst.compute_trajectory(50.0, 0) is None


def test_constraint_params_tuples(gpu, example):
    fx, path, pc_vel, pc_acc = example
    grid = fx["n100_grid"]
    a, b, c, F, g, ub, xb = pc_acc.compute_constraint_params(path, grid)
    assert_same(a, fx["n100_acc_a"], "a"); assert_same(b, fx["n100_acc_b"], "b"); assert_same(c, fx["n100_acc_c"], "c")
    assert_same(F, fx["n100_acc_F"], "F"); assert_same(g, fx["n100_acc_g"], "g")
    assert ub is None and xb is None
    out = pc_vel.compute_constraint_params(path, grid)
    assert all(o is None for o in out[:6])
    assert_same(out[6], fx["n100_xbound"], "xbound")
    # Collocation: a == q'(s), b == q''(s)  (tests/tests/constraint/test_joint_acceleration.py:68-92)
    pc = ta.constraint.JointAccelerationConstraint(fx["alim"][0], discretization_scheme=0)
    a, b, c, F, g, _, _ = pc.compute_constraint_params(path, grid)
    assert_same(a, path(grid, 1), "colloc a"); assert_same(b, path(grid, 2), "colloc b")
    assert np.array_equal(F, np.vstack([np.eye(7), -np.eye(7)]))
    with pytest.raises(ValueError):
        ta.constraint.JointAccelerationConstraint([1.0, 1.0]).compute_constraint_params(path, grid)


@pytest.mark.parametrize("lp1d", [0, 1])
def test_solve_stagewise_optim_sequence(gpu, example, lp1d):
    """The single-LP entry with its stateful warm start, same call sequence as the reference object."""
    fx, path, pc_vel, pc_acc = example
    w = ta.solverwrapper.hipSeidelWrapper([pc_vel, pc_acc], path, fx["n100_grid"], solve_lp1d=lp1d)
    q, r = fx["stagewise_q_lp1d%d" % lp1d], fx["stagewise_r_lp1d%d" % lp1d]
    for row, want in zip(q, r):
        got = w.solve_stagewise_optim(int(row[0]), None, row[1:3], *row[3:7])
        assert_same(got, want, "stagewise")
    # infeasible instance -> [nan, nan]  (test_basic_can_linear.py:167-200)
    assert np.all(np.isnan(w.solve_stagewise_optim(0, None, np.r_[0.0, 1.0], 1.0, 0.5, np.nan, np.nan)))
    assert len(w.params) == 2 and w.params[0][6].shape == (101, 2)


def test_reference_objects_are_accepted(gpu, example):
    """Duck typing: any path exposing .cspl and constraints exposing .vlim/.alim work."""
    fx, path, pc_vel, pc_acc = example

    class RefLikeAcc(object):  # shaped like toppra.constraint.JointAccelerationConstraint
        alim = fx["alim"][0]

        class _E(object):
            value = 1
        def get_constraint_type(self):
            class T(object):
                value = 0
            return T()
        def get_discretization_type(self):
            return self._E()
        def get_dof(self):
            return 7
    inst = ta.algorithm.TOPPRA([pc_vel, RefLikeAcc()], path, gridpoints=fx["n100_grid"])
    _, sd, _ = inst.compute_parameterization(0, 0)
    assert_same(sd, fx["n100_sd"], "sd_vec")


def test_batch_toppra_and_torch_device_path(gpu):
    import torch
    fx = golden("batch_d7_N200")
    bt = ta.algorithm.BatchTOPPRA(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"])
    host = bt.compute_parameterization()
    assert_same(host["sd"], fx["sd"], "sd")
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(fx[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")}
    bd = ta.algorithm.BatchTOPPRA(t["coef"], t["breaks"], t["grid"], t["vlim"], t["alim"])
    out = bd.compute_parameterization()
    assert out["sd2"].is_cuda
    assert_same(out["sd"].cpu().numpy(), fx["sd"], "sd (device pointers)")
    assert_same(out["K"].cpu().numpy(), fx["K"], "K (device pointers)")
    assert bt.return_codes(host["status"])[0] == ParameterizationReturnCode.Ok
    # per-trajectory grids == shared grid
    grid2 = np.tile(fx["grid"], (fx["coef"].shape[0], 1))
    out2 = ta.batch.solve_batch(fx["coef"], fx["breaks"], grid2, fx["vlim"], fx["alim"], want_sd=True)
    assert_same(out2["sd"], fx["sd"], "sd (per-trajectory grid)")
    br2 = np.tile(fx["breaks"], (fx["coef"].shape[0], 1))
    out3 = ta.batch.solve_batch(fx["coef"], br2, grid2, fx["vlim"], fx["alim"], want_sd=True, variant=1)
    assert_same(out3["sd"], fx["sd"], "sd (per-trajectory breaks)")


@pytest.mark.parametrize("scheme", [0, 1])
def test_robust_constraint_params(gpu, example, scheme):
    """RobustLinearConstraint.compute_constraint_params vs the reference's output."""
    fx, path, pc_vel, pc_acc = example
    rc = ta.constraint.RobustLinearConstraint(ta.constraint.JointAccelerationConstraint(fx["alim"][0]),
                                              [1e-3, 5e-2, 9e-3], scheme)
    a, b, c, P, ub, xb = rc.compute_constraint_params(path, fx["n100_grid"])
    assert_same(a, fx["robust%d_a" % scheme], "a"); assert_same(b, fx["robust%d_b" % scheme], "b")
    assert_same(c, fx["robust%d_c" % scheme], "c"); assert_same(P, fx["robust%d_P" % scheme], "P")
    assert ub is None and xb is None
    assert rc.get_constraint_type() == ta.constraint.ConstraintType.CanonicalConic
    with pytest.raises(AssertionError):  # as in the reference: seidel cannot take conic constraints
        ta.algorithm.TOPPRA([pc_vel, rc], path, gridpoints=fx["n100_grid"], solver_wrapper="seidel")


def test_K_is_optional_and_inputs_are_validated(gpu):
    """tpr_result.K == NULL keeps the controllable sets in a workspace (same sd2 / u / status bits);
    mistyped or misplaced device tensors are refused instead of being read as raw fp64 pointers."""
    import torch
    data = batch.make_synthetic_batch(300, 5, 40, seed=11)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    full = batch.solve_batch(*args)
    for variant in (0, 2, 3, 4):
        lean = batch.solve_batch(*args, want_K=False, variant=variant)
        assert "K" not in lean
        for k in ("sd2", "u", "status"):
            assert np.array_equal(lean[k], full[k], equal_nan=True), (variant, k)
        leaner = batch.solve_batch(*args, want_K=False, want_u=False, variant=variant)  # what retiming reads: sd2 only
        assert "u" not in leaner and "K" not in leaner
        for k in ("sd2", "status"):
            assert np.array_equal(leaner[k], full[k], equal_nan=True), (variant, k)
    scalar = batch.solve_batch(*args, sd_start=0.0, sd_end=0.05)           # scalars broadcast to [B]
    vec = batch.solve_batch(*args, sd_start=np.zeros(300), sd_end=np.full(300, 0.05))
    assert np.array_equal(scalar["sd2"], vec["sd2"], equal_nan=True)
    dev = torch.device("cuda", 0)
    dv = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in args]
    torch.cuda.set_device(0)
    got = batch.solve_batch(*dv)
    assert np.array_equal(got["sd2"].cpu().numpy(), full["sd2"], equal_nan=True)
    with pytest.raises(ValueError):
        batch.solve_batch(dv[0].float(), *dv[1:])                            # fp32 tensor
    with pytest.raises(ValueError):
        batch.solve_batch(dv[0], dv[1], dv[2], dv[3].cpu(), dv[4])           # host tensor mixed in
    with pytest.raises(ValueError):
        batch.solve_batch(*dv, sd_end=torch.zeros(7, dtype=torch.float64, device=dev))


def test_compute_reachable_sets_drop_in(gpu, oracle):
    """ReachabilityAlgorithm.compute_reachable_sets on the drop-in class vs the oracle (itself pinned to the
    reference live and through tests/golden/reach_*.npz), and the batch method on the reference fixture."""
    rng = np.random.default_rng(12)
    way = rng.standard_normal((5, 4))
    path = ta.SplineInterpolator(np.linspace(0, 1, 5), way)
    vl = np.stack([-np.full(4, 15.0), np.full(4, 15.0)], 1)
    al = np.stack([-np.full(4, 11.0), np.full(4, 11.0)], 1)
    grid = np.linspace(0, 1, 51)
    inst = ta.algorithm.TOPPRA([ta.constraint.JointVelocityConstraint(vl), ta.constraint.JointAccelerationConstraint(al)],
                               path, gridpoints=grid)
    L = inst.compute_reachable_sets(0.0, 0.2)
    w = oracle.Wrapper(path.cspl.c, path.cspl.x, grid, vl, al)
    wantL, wantX = w.compute_reachable_sets(0.0, 0.2)
    assert_same(L, wantL, "L")
    assert_same(inst.problem_data.X, wantX, "X")
    fx = golden("reach_d5_N60")
    binst = ta.algorithm.BatchTOPPRA(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"])
    assert_same(binst.compute_reachable_sets(fx["sdmin"], fx["sdmax"]), fx["L"], "batch L")


def _reuse_instance(fx, b):
    from tests.helpers import path_from_tables
    path = path_from_tables(fx["coef"][b], fx["breaks"])
    scheme = ta.constraint.DiscretizationType(int(fx["scheme"][b]))
    cons = [ta.constraint.JointVelocityConstraint(fx["vlim"][b]),
            ta.constraint.JointAccelerationConstraint(fx["alim"][b], discretization_scheme=scheme)]
    return ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"])


def test_passes_chained_on_one_instance_match_the_reference(gpu):
    """examples/plot_kinematics.py:48,72 run several passes on ONE algorithm instance; the reference's wrapper object
    carries its warm-start state (active_c_up / active_c_down) from pass to pass, and the later passes pivot from
    there.  tests/golden/reuse_d7_N100 holds compute_parameterization -> compute_feasible_sets ->
    compute_controllable_sets -> compute_parameterization on one reference instance per trajectory (in 9 / 5 of the
    24 the carried state changes bits of X / K against a fresh instance): reproduced bit for bit through the class,
    whose wrapper threads the state through every pass (tpr_problem.active)."""
    fx = golden("reuse_d7_N100")
    B = fx["coef"].shape[0]
    differs = 0
    for b in range(B):
        inst = _reuse_instance(fx, b)
        sd0, sd1 = float(fx["sd_start"][b]), float(fx["sd_end"][b])
        st_code = {0: "Ok", 1: "FailUncontrollable", 2: "ErrUnknown"}

        def param(tag):
            sdd, sd, _, K = inst.compute_parameterization(np.float64(sd0), np.float64(sd1), return_data=True)
            assert inst.problem_data.return_code.name == st_code[int(fx["status" + tag][b])], (b, tag)
            assert_same(K, fx["K" + tag][b], "K%s[%d]" % (tag, b))
            if sd is not None:
                assert_same(sd, fx["sd" + tag][b], "sd%s[%d]" % (tag, b))
                assert_same(sdd, fx["u" + tag][b], "u%s[%d]" % (tag, b))
        param("1")
        assert_same(inst.compute_feasible_sets(), fx["X"][b], "X[%d]" % b)
        assert_same(inst.compute_controllable_sets(np.float64(fx["sdmin"][b]), np.float64(fx["sdmax"][b])), fx["K2"][b], "K2[%d]" % b)
        param("3")
        # ... and a FRESH instance gives the fresh-instance bits
        fresh = _reuse_instance(fx, b)
        assert_same(fresh.compute_feasible_sets(), fx["X_fresh"][b], "X_fresh[%d]" % b)
        differs += not np.array_equal(fx["X"][b], fx["X_fresh"][b], equal_nan=True)
    assert differs >= 5  # the fixture does exercise the carried state


def test_warm_start_state_through_the_batch_entries(gpu, oracle):
    """tpr_problem.active on the batch entries: the same chain for a whole batch at once, against the oracle's
    wrapper objects (pinned to the reference's by tests/test_oracle_vs_reference.py), states included."""
    fx = golden("reuse_d7_N100")
    keep = np.flatnonzero(fx["scheme"] == 1)
    coef, vlim, alim = fx["coef"][keep], fx["vlim"][keep], fx["alim"][keep]
    B = len(keep)
    active = np.zeros((B, 4), dtype=np.int32)
    out = batch.solve_batch(coef, fx["breaks"], fx["grid"], vlim, alim, fx["sd_start"][keep], fx["sd_end"][keep], active=active)
    X = batch.feasible_sets_batch(coef, fx["breaks"], fx["grid"], vlim, alim, active=active)
    K2 = batch.controllable_sets_batch(coef, fx["breaks"], fx["grid"], vlim, alim, fx["sdmin"][keep], fx["sdmax"][keep], active=active)
    assert_same(out["K"], fx["K1"][keep], "K1")
    assert_same(X, fx["X"][keep], "X")
    assert_same(K2, fx["K2"][keep], "K2")
    for j, b in enumerate(keep):
        w = oracle.Wrapper(fx["coef"][b], fx["breaks"], fx["grid"], fx["vlim"][b], fx["alim"][b])
        w.compute_parameterization(float(fx["sd_start"][b]), float(fx["sd_end"][b]))
        w.compute_feasible_sets()
        w.compute_controllable_sets(float(fx["sdmin"][b]), float(fx["sdmax"][b]))
        assert np.array_equal(w.active(), active[j]), (b, w.active(), active[j])


def test_boundary_velocities_squared_like_the_reference(gpu):
    """`sd ** 2` on Python floats is libm's pow(sd, 2.0), one ulp off sd * sd for ~0.08 % of the doubles; the
    reference squares its boundary velocities that way (reachability_algorithm.py:226,:262).  Every boundary
    velocity of tests/golden/pow_boundary_d3_N40 is such a value: the class squares with the same expression and
    hands x = sd^2 down (TPR_BOUNDARY_SQUARED), so K, sd and u are the reference's bits; squaring on the device
    (what the array entries do, correctly for numpy inputs) would be off in K[N] of every trajectory."""
    from tests.helpers import path_from_tables
    fx = golden("pow_boundary_d3_N40")
    B, N = fx["coef"].shape[0], len(fx["grid"]) - 1
    assert all(float(v) ** 2 != float(v) * float(v) for v in fx["sd_end"])
    for b in range(B):
        path = path_from_tables(fx["coef"][b], fx["breaks"])
        cons = [ta.constraint.JointVelocityConstraint(fx["vlim"][b]), ta.constraint.JointAccelerationConstraint(fx["alim"][b])]
        inst = ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"])
        sdd, sd, _, K = inst.compute_parameterization(float(fx["sd_start"][b]), float(fx["sd_end"][b]), return_data=True)
        assert_same(K, fx["K"][b], "K[%d]" % b)
        if int(fx["status"][b]) == 0:
            assert_same(sd, fx["sd"][b], "sd[%d]" % b)
            assert_same(sdd, fx["u"][b], "u[%d]" % b)
        else:
            assert sd is None
        fresh = ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"])
        assert_same(fresh.compute_controllable_sets(float(fx["sdmin"][b]), float(fx["sdmax"][b])), fx["Kc"][b], "Kc[%d]" % b)
    dev = batch.solve_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["sd_start"], fx["sd_end"])
    assert (dev["K"][:, N, 0] != fx["K"][:, N, 0]).all()  # sd * sd on the device: the other rounding of the same square


def test_warm_start_state_on_long_grids_and_forced_variants(gpu, oracle):
    """tpr_problem.active where kernel family 4 cannot take the problem (N > 1480: its per-trajectory tables no longer fit
    the LDS): the generic lane kernel (family 1) carries the state instead -- the chain compute_parameterization ->
    compute_feasible_sets -> compute_controllable_sets -> compute_parameterization on one object, against the oracle's
    wrapper objects, states included; forcing a family that does not maintain the state is refused, not ignored."""
    from toppra_amd import _capi
    B, d, N = 6, 3, 1600
    data = batch.make_synthetic_batch(B, d, N, seed=61)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    active = np.zeros((B, 4), dtype=np.int32)
    sd1 = np.array([0.0, 0.1, 0.0, 0.2, 0.0, 0.05])
    out1 = batch.solve_batch(*args, None, sd1, active=active)
    X = batch.feasible_sets_batch(*args, active=active)
    K2 = batch.controllable_sets_batch(*args, np.zeros(B), 0.3 * np.ones(B), active=active)
    out3 = batch.solve_batch(*args, None, sd1, active=active)
    changed = 0
    for b in range(B):
        w = oracle.Wrapper(data["coef"][b], data["breaks"], data["grid"], data["vlim"][b], data["alim"][b])
        st, sdd, sd, xs, K = w.compute_parameterization(0.0, float(sd1[b]))
        assert st == out1["status"][b] and np.array_equal(K, out1["K"][b], equal_nan=True), b
        assert np.array_equal(w.compute_feasible_sets(), X[b], equal_nan=True), b
        assert np.array_equal(w.compute_controllable_sets(0.0, 0.3), K2[b], equal_nan=True), b
        st, sdd, sd, xs, K = w.compute_parameterization(0.0, float(sd1[b]))
        assert st == out3["status"][b] and np.array_equal(K, out3["K"][b], equal_nan=True), b
        if st == 0:
            assert np.array_equal(xs, out3["sd2"][b]) and np.array_equal(sdd, out3["u"][b]), b
        assert np.array_equal(w.active(), active[b]), (b, w.active(), active[b])
        changed += bool(active[b].any())
    assert changed == B
    small = batch.make_synthetic_batch(4, 3, 40, seed=62)
    sargs = (small["coef"], small["breaks"], small["grid"], small["vlim"], small["alim"])
    for variant in (2, 3):
        with pytest.raises(_capi.ToppraHipError):
            batch.solve_batch(*sargs, active=np.zeros((4, 4), dtype=np.int32), variant=variant)

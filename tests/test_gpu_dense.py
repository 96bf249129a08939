"""The dense-row entries (tpr_*_dense_batch: the seidel path for ANY canonical-linear constraint list, rows given as the
arrays the reference's seidelWrapper builds) against the reference's own outputs for torque / second-order constraints
(tests/golden/dense_*.npz, tools/make_golden.py), against the oracle on random dense problems, and against the fused
kernels on the rows of the standard velocity + acceleration problem."""
import numpy as np
import pytest

import toppra_amd as ta
from tests.helpers import assert_same, dense_constraints, dense_fixtures, golden
from toppra_amd import batch
from toppra_amd.solverwrapper import dense_rows, hipDenseSeidelWrapper

pytestmark = pytest.mark.gpu


def _rows(fx):
    return fx["a"], fx["b"], fx["c"], fx["low"], fx["high"], fx["deltas"]


@pytest.mark.parametrize("name", dense_fixtures())
def test_dense_fixture(gpu, name):
    """JointTorqueConstraint / SecondOrderConstraint problems solved by the REFERENCE (seidel): K, sd, u, return codes,
    feasible sets and controllable sets from the fixture's dense rows, bit for bit."""
    fx = golden(name)
    got = batch.solve_dense_batch(*_rows(fx), fx["sd_start"], fx["sd_end"], want_sd=True)
    assert np.array_equal(got["status"], fx["status"])
    assert_same(got["K"], fx["K"], "K")
    assert_same(got["sd"], fx["sd"], "sd")
    assert_same(got["u"], fx["u"], "u")
    ok = fx["status"] == 0
    assert_same(got["sd2"][ok], fx["sd"][ok] ** 2, "sd2", atol=1e-8)
    assert_same(batch.feasible_sets_dense_batch(*_rows(fx)), fx["X"], "X")
    B = fx["a"].shape[0]
    Kc = batch.controllable_sets_dense_batch(*_rows(fx), np.full(B, float(fx["sdmin_c"])), np.full(B, float(fx["sdmax_c"])))
    assert_same(Kc, fx["Kc"], "controllable sets")
    # reachable sets from an interval and from a point (sdmin == sdmax: the first stage takes the 1-variable path, whose
    # active index seeds the later warm starts)
    L, X = batch.reachable_sets_dense_batch(*_rows(fx), np.zeros(B), np.full(B, 0.3), want_X=True)
    assert_same(L, fx["L"], "reachable sets")
    assert_same(X, fx["X"], "X of the reachable-set pass")
    assert_same(batch.reachable_sets_dense_batch(*_rows(fx), np.full(B, 0.1), np.full(B, 0.1)), fx["L_point"], "reachable sets from a point")


@pytest.mark.parametrize("name", dense_fixtures())
def test_dense_fixture_through_the_drop_in_classes(gpu, name):
    """The same problems through toppra_amd's own classes -- SplineInterpolator, JointTorqueConstraint /
    SecondOrderConstraint (the user's inverse dynamics evaluated on the host, as in the reference), TOPPRA -- i.e. what a
    user of the reference would write: the algorithm picks hipDenseSeidelWrapper for constraint lists the fused kernels
    do not regenerate, and returns the reference's bits."""
    fx = golden(name)
    for b in range(fx["a"].shape[0]):
        path = ta.SplineInterpolator(fx["knots"], fx["way"][b])
        cons = dense_constraints(fx, b, ta.constraint)
        rows = dense_rows(cons, path, fx["grid"])
        for k in ("a", "b", "c", "low", "high"):
            assert_same(rows[k], fx[k][b], "%s[%d]" % (k, b))
        inst = ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"])
        assert isinstance(inst.solver_wrapper, hipDenseSeidelWrapper)
        # (numpy scalars, as the fixture script passed them: `sd ** 2` is numpy's square there, libm pow for Python floats)
        sdd, sd, _, K = inst.compute_parameterization(fx["sd_start"][b], fx["sd_end"][b], return_data=True)
        assert_same(K, fx["K"][b], "K[%d]" % b)
        if fx["status"][b] == 0:
            assert_same(sd, fx["sd"][b], "sd[%d]" % b)
            assert_same(sdd, fx["u"][b], "u[%d]" % b)
            traj = ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"]).compute_trajectory(fx["sd_start"][b], fx["sd_end"][b])
            assert traj is not None and traj.duration > 0
        else:
            assert sd is None and sdd is None
        assert_same(ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"]).compute_feasible_sets(), fx["X"][b], "X[%d]" % b)
        assert_same(ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"]).compute_controllable_sets(
            float(fx["sdmin_c"]), float(fx["sdmax_c"])), fx["Kc"][b], "Kc[%d]" % b)
        assert_same(ta.algorithm.TOPPRA(cons, path, gridpoints=fx["grid"]).compute_reachable_sets(0.0, 0.3), fx["L"][b], "L[%d]" % b)


@pytest.mark.parametrize("B,N,nC,seed", [(64, 30, 2, 1), (200, 25, 7, 2), (96, 40, 34, 3), (64, 20, 35, 4), (40, 16, 66, 5), (33, 1, 12, 6), (24, 12, 67, 7), (16, 10, 122, 8),
                                        (3000, 40, 30, 9), (1500, 30, 50, 10)])
def test_random_dense_problems_vs_oracle(gpu, oracle, B, N, nC, seed):
    """Random dense rows -- any row count up to the 122 the slot layouts hold (34 / 35 and 66 / 67: the switches from 8 to 16 to 32
    lanes per trajectory), rows of mixed orientation, boxes on u, infeasible stages, failing forward scans -- against the oracle's
    seidelWrapper on the same arrays: K, sd2, u, return codes, feasible sets, bit for bit."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, size=(B, N + 1, nC))
    a, b = np.cos(ang), 0.3 * np.sin(ang)
    c = -(0.5 + 2.0 * rng.random((B, N + 1, nC)))           # the origin strictly inside every row
    c[rng.random((B, N + 1, nC)) < 0.4 / ((N + 1) * nC)] = 0.3  # ... except in a third of the trajectories: infeasible or tight stages
    a[:, :, :2] = b[:, :, :2] = c[:, :, :2] = 0.0            # the reserved x_next rows
    low = np.stack([np.full((B, N + 1), -1e8), np.zeros((B, N + 1))], axis=-1)
    high = np.stack([np.full((B, N + 1), 1e8), 2.0 + rng.random((B, N + 1))], axis=-1)
    tight = rng.random(B) < 0.3                              # a box on u for some trajectories (ubound)
    low[tight, :, 0], high[tight, :, 0] = -3.0, 3.0
    deltas = 0.02 + 0.03 * rng.random((B, N))
    sd0, sd1 = 0.5 * rng.random(B), 0.5 * rng.random(B)
    sd0[::7] = 5.0                                           # outside the controllable set (x <= 3)
    got = batch.solve_dense_batch(a, b, c, low, high, deltas, sd0, sd1, want_sd=True)
    want = oracle.solve_dense_batch(a, b, c, low, high, deltas, sd0, sd1, want_X=True)
    assert np.array_equal(got["status"], want["status"]) and len(set(want["status"])) >= 2
    assert_same(got["K"], want["K"], "K")
    done = want["status"] == 0
    for k in ("sd2", "sd", "u"):
        assert_same(got[k][done], want[k][done], k)
    assert np.isnan(got["sd2"][want["status"] == 1]).all()
    assert_same(batch.feasible_sets_dense_batch(a, b, c, low, high, deltas), want["X"], "X")


@pytest.mark.parametrize("B,d,N,interp,vel", [(4096, 7, 200, True, True), (300, 7, 60, False, True), (256, 3, 50, True, False),
                                              (64, 9, 40, True, True), (32, 16, 25, True, True), (200, 5, 3, True, True)])
def test_dense_rows_of_the_standard_problem_give_the_fused_kernels_bits(gpu, B, d, N, interp, vel):
    """tpr_constraint_params_batch writes the rows of the velocity + acceleration problem as seidelWrapper would; fed back
    as dense arrays they must give what the fused kernels (which never materialise them, and answer most LPs from
    certificates) give: parameterization, controllable sets from an interval, feasible sets -- every bit."""
    rng = np.random.default_rng(B + d)
    data = batch.make_synthetic_batch(B, d, N, seed=100 + d + N)
    vlim = data["vlim"] if vel else None
    sd0 = 0.3 * rng.random(B) * (rng.random(B) < 0.5)
    sd1 = 0.3 * rng.random(B) * (rng.random(B) < 0.5)
    args = (data["coef"], data["breaks"], data["grid"], vlim, data["alim"])
    rows = batch.constraint_params_batch(*args, interp)
    dense = (rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], np.diff(data["grid"]))
    ref = batch.solve_batch(*args, sd0, sd1, interp, want_sd=True)
    got = batch.solve_dense_batch(*dense, sd0, sd1, want_sd=True)
    for k in ("K", "sd2", "sd", "u"):
        assert_same(got[k], ref[k], k)
    assert np.array_equal(got["status"], ref["status"])
    assert_same(batch.controllable_sets_dense_batch(*dense, 0.1 * sd1, sd1 + 0.2),
                batch.controllable_sets_batch(*args, 0.1 * sd1, sd1 + 0.2, interp), "controllable sets")
    assert_same(batch.feasible_sets_dense_batch(*dense), batch.feasible_sets_batch(*args, interp), "feasible sets")


def test_dense_entries_on_device_tensors_and_bad_arguments(gpu):
    import torch
    fx = golden("dense_torque_d5_N40_collocation")
    dev = torch.device("cuda", 0)
    rows = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in _rows(fx)]
    got = batch.solve_dense_batch(*rows, torch.from_numpy(fx["sd_start"]).to(dev), torch.from_numpy(fx["sd_end"]).to(dev), want_sd=True)
    assert_same(got["sd"].cpu().numpy(), fx["sd"], "sd (device tensors)")
    assert_same(batch.feasible_sets_dense_batch(*rows).cpu().numpy(), fx["X"], "X (device tensors)")
    a = np.zeros((2, 5, 123))
    with pytest.raises(ValueError):
        batch.solve_dense_batch(a, a, a, np.zeros((2, 5, 2)), np.ones((2, 5, 2)), np.ones(4))
    with pytest.raises(ValueError):
        batch.solve_dense_batch(a[:, :, :9], a[:, :, :8], a[:, :, :9], np.zeros((2, 5, 2)), np.ones((2, 5, 2)), np.ones(4))
    with pytest.raises(ValueError):
        batch.solve_dense_batch(a[:, :, :9], a[:, :, :9], a[:, :, :9], np.zeros((2, 5, 2)), np.ones((2, 5, 2)), np.ones(3))


def test_varying_velocity_limits_with_a_constant_function_give_the_fused_path(gpu, oracle):
    """JointVelocityConstraintVarying is evaluated on the host (a Python callback per gridpoint) and sends the problem to
    the dense-row entries.  With a constant function it states the fused problem: same bounds (the fp32 rounding of the
    reference's velocity bound reproduced on the host as on the device), same K / sd / u through two unrelated paths --
    host rows + full iteration vs rows regenerated on the GPU + certificates.  With a varying function: the oracle."""
    rng = np.random.default_rng(11)
    knots, grid = np.linspace(0, 1, 5), np.linspace(0, 1, 81)
    for d in (2, 5, 7):
        way = rng.standard_normal((5, d))
        vlim = np.stack([-2 - 3 * rng.random(d), 2 + 3 * rng.random(d)], axis=1)   # tight: the velocity bound is active
        alim = np.stack([-10 - 2 * rng.random(d), 10 + 2 * rng.random(d)], axis=1)
        path = ta.SplineInterpolator(knots, way)
        acc = ta.constraint.JointAccelerationConstraint(alim)
        fused = ta.algorithm.TOPPRA([ta.constraint.JointVelocityConstraint(vlim), acc], path, gridpoints=grid)
        dense = ta.algorithm.TOPPRA([ta.constraint.JointVelocityConstraintVarying(lambda s: vlim), acc], path, gridpoints=grid)
        assert isinstance(dense.solver_wrapper, hipDenseSeidelWrapper) and not isinstance(fused.solver_wrapper, hipDenseSeidelWrapper)
        want, got = fused.compute_parameterization(0, 0, return_data=True), dense.compute_parameterization(0, 0, return_data=True)
        for k, name in ((0, "u"), (1, "sd"), (3, "K")):
            assert_same(got[k], want[k], "%s (d = %d)" % (name, d))
        assert_same(dense.compute_feasible_sets(), ta.algorithm.TOPPRA([ta.constraint.JointVelocityConstraint(vlim), acc], path,
                                                                      gridpoints=grid).compute_feasible_sets(), "X")
        wavy = ta.algorithm.TOPPRA([ta.constraint.JointVelocityConstraintVarying(lambda s: vlim * (1 + 0.5 * np.sin(9 * s))), acc], path,
                                   gridpoints=grid)
        sdd, sd, _, K = wavy.compute_parameterization(0, 0, return_data=True)
        rows = wavy.solver_wrapper._rows
        w = oracle.DenseWrapper(*[r[0] for r in rows[:5]], rows[5])
        st, osdd, osd, oxs, oK = w.compute_parameterization(0.0, 0.0)
        assert st == 0
        assert_same(K, oK, "K (varying)")
        assert_same(sd, osd, "sd (varying)")
        assert_same(sdd, osdd, "u (varying)")


def test_dense_wrapper_single_lp_entry(gpu, oracle):
    """hipDenseSeidelWrapper.solve_stagewise_optim (the reference's per-stage contract, cy_seidel_solverwrapper.pyx:549-697)
    against the oracle's on the same rows: a backward sweep of 2-D LPs with the warm-start state carried from call to
    call, absent (NaN) bounds, the last stage, and the 1-variable path of the forward step -- every [u, x] bit for bit."""
    fx = golden("dense_vel_acc_second_d6_N50")
    b = int(np.flatnonzero(fx["status"] == 0)[0])
    path = ta.SplineInterpolator(fx["knots"], fx["way"][b])
    w = hipDenseSeidelWrapper(dense_constraints(fx, b, ta.constraint), path, fx["grid"], solve_lp1d=1)
    o = oracle.DenseWrapper(fx["a"][b], fx["b"][b], fx["c"][b], fx["low"][b], fx["high"][b], fx["deltas"], solve_lp1d=1)
    N = w.get_no_stages()
    assert w.get_no_vars() == 2 and N == 50 and np.array_equal(w.get_deltas(), fx["deltas"])
    nan = float("nan")
    K = fx["K"][b]
    calls = [(N, [1e-9, -1.0], nan, nan, nan, nan), (N, [-1e-9, 1.0], 0.0, 1e4, nan, nan)]
    for i in range(N - 1, 30, -1):                      # _one_step: upper then lower bound of the controllable set
        calls += [(i, [1e-9, -1.0], nan, nan, K[i + 1, 0], K[i + 1, 1]), (i, [-1e-9, 1.0], nan, nan, K[i + 1, 0], K[i + 1, 1])]
    for i in range(0, 8):                               # the forward step: x_min == x_max -> the 1-variable LP
        x = float(fx["sd"][b, i]) ** 2
        calls.append((i, [-2 * fx["deltas"][i], -1.0], x, x, K[i + 1, 0], K[i + 1, 1]))
    calls.append((3, [0.3, 0.7], 0.1, 0.2, nan, 5.0))    # a generic objective, one-sided x_next bound
    for i, g, x0, x1, n0, n1 in calls:
        got = w.solve_stagewise_optim(i, None, np.array(g), x0, x1, n0, n1)
        want = o.solve_stagewise_optim(i, None, np.array(g), x0, x1, n0, n1)
        assert_same(np.asarray(got), np.asarray(want), "stage %d g %s" % (i, g))


@pytest.mark.parametrize("name", dense_fixtures())
def test_dense_fixture_desired_duration(gpu, name):
    """TOPPRAsd of the REFERENCE on torque / second-order constraint lists: tpr_solve_desired_duration_dense_batch on the
    fixture's rows, and toppra_amd.algorithm.TOPPRAsd on toppra_amd's own constraint classes -- K, sd, u, return codes bit
    for bit (desired durations below the fastest, in range -- the bisection --, above the slowest; uncontrollable starts)."""
    fx = golden(name)
    got = batch.solve_desired_duration_dense_batch(*_rows(fx), fx["sd_desired"], fx["sd_start"], fx["sd_end"])
    assert np.array_equal(got["status"], fx["sd_status"])
    assert_same(got["K"], fx["sd_K"], "K")
    assert_same(got["sd"], fx["sd_sd"], "sd")
    assert_same(got["u"], fx["sd_u"], "u")
    ok = fx["sd_status"] == 0
    assert np.all((got["alpha"][ok] >= 0) & (got["alpha"][ok] <= 1)) and np.any((got["alpha"][ok] > 0) & (got["alpha"][ok] < 1))
    for b in range(fx["a"].shape[0]):
        inst = ta.algorithm.TOPPRAsd(dense_constraints(fx, b, ta.constraint), ta.SplineInterpolator(fx["knots"], fx["way"][b]),
                                     gridpoints=fx["grid"])
        inst.set_desired_duration(float(fx["sd_desired"][b]))
        sdd, sd, _, K = inst.compute_parameterization(fx["sd_start"][b], fx["sd_end"][b], return_data=True)
        assert_same(K, fx["sd_K"][b], "K[%d]" % b)
        if fx["sd_status"][b] != 1:  # (2 = a NaN in the blended profile: arrays with NaNs, as the reference returns them)
            assert_same(sd, fx["sd_sd"][b], "sd[%d]" % b)
            assert_same(sdd, fx["sd_u"][b], "u[%d]" % b)
        else:
            assert sd is None


def test_dense_desired_duration_vs_the_fused_path(gpu):
    """TOPPRAsd on the standard problem's own rows (dense) against the fused TOPPRAsd kernels: every output bit, blend
    factors included, on a batch with desired durations below, inside and far above the achievable range and boundary velocities."""
    B, d, N = 600, 6, 80
    rng = np.random.default_rng(4)
    data = batch.make_synthetic_batch(B, d, N, seed=9)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    sd0 = 0.2 * rng.random(B) * (rng.random(B) < 0.4)
    desired = np.choose(np.arange(B) % 4, [0.3, 1.5, 3.0, 1e5]) * (1 + 0.2 * rng.random(B))
    ref = batch.solve_desired_duration_batch(*args, desired, sd0, None)
    rows = batch.constraint_params_batch(*args)
    got = batch.solve_desired_duration_dense_batch(rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], np.diff(data["grid"]),
                                                   desired, sd0, None)
    for k in ("K", "sd2", "sd", "u", "alpha"):
        assert_same(got[k], ref[k], k)
    assert np.array_equal(got["status"], ref["status"])
    assert np.any((ref["alpha"] > 0) & (ref["alpha"] < 1)) and np.any(ref["alpha"] == 1.0)


def test_dense_passes_chained_on_one_instance(gpu):
    """tests/golden/dense_reuse_d5_N60: compute_parameterization -> compute_feasible_sets -> compute_controllable_sets ->
    compute_reachable_sets -> compute_parameterization on ONE reference instance with a torque constraint.  The wrapper
    object's warm-start state (tpr_dense_problem.active: written by the 2-D LPs and by the 1-variable path of the forward
    scan) is carried through toppra_amd's instance the same way: every pass returns the reference's bits (a fresh object
    per pass would not, in 7 of the 16 trajectories) -- through the drop-in classes, and through the batch entries with
    one `active` array for the whole batch."""
    fx = golden("dense_reuse_d5_N60")
    B = fx["a"].shape[0]
    for b in range(B):
        inst = ta.algorithm.TOPPRA(dense_constraints(fx, b, ta.constraint), ta.SplineInterpolator(fx["knots"], fx["way"][b]),
                                   gridpoints=fx["grid"])
        sdd, sd, _, K = inst.compute_parameterization(fx["sd_start"][b], fx["sd_end"][b], return_data=True)
        assert_same(K, fx["K"][b], "K[%d]" % b); assert_same(sd, fx["sd"][b], "sd[%d]" % b); assert_same(sdd, fx["u"][b], "u[%d]" % b)
        assert_same(inst.compute_feasible_sets(), fx["X"][b], "X[%d] after the parameterization" % b)
        assert_same(inst.compute_controllable_sets(0.05, 0.4), fx["Kc"][b], "Kc[%d] after that" % b)
        assert_same(inst.compute_reachable_sets(0.0, 0.3), fx["L"][b], "L[%d] after that" % b)
        sdd, sd, _, K = inst.compute_parameterization(fx["sd_start"][b], fx["sd_end"][b], return_data=True)
        assert_same(K, fx["K2"][b], "K[%d], second parameterization" % b)
        assert_same(sd, fx["sd2nd"][b], "sd[%d], second parameterization" % b)
        assert_same(sdd, fx["u2nd"][b], "u[%d], second parameterization" % b)
    active = np.zeros((B, 4), dtype=np.int32)
    got = batch.solve_dense_batch(*_rows(fx), fx["sd_start"], fx["sd_end"], want_sd=True, active=active)
    assert_same(got["sd"], fx["sd"], "sd (batch)")
    assert active.any()
    assert_same(batch.feasible_sets_dense_batch(*_rows(fx), active=active), fx["X"], "X (batch, carried state)")
    assert_same(batch.controllable_sets_dense_batch(*_rows(fx), np.full(B, 0.05), np.full(B, 0.4), active=active), fx["Kc"], "Kc (batch)")
    assert_same(batch.reachable_sets_dense_batch(*_rows(fx), np.zeros(B), np.full(B, 0.3), active=active), fx["L"], "L (batch)")
    again = batch.solve_dense_batch(*_rows(fx), fx["sd_start"], fx["sd_end"], want_sd=True, active=active)
    assert_same(again["K"], fx["K2"], "K (batch, second parameterization)")
    assert_same(again["sd"], fx["sd2nd"], "sd (batch, second parameterization)")
    fresh = batch.feasible_sets_dense_batch(*_rows(fx))
    assert not np.array_equal(fresh, fx["X"], equal_nan=True)   # the state matters: a fresh object returns other bits


def test_dense_rows_at_the_headline_shape(gpu):
    """65 536 x 7 x 200 (BASELINE's headline batch), device-resident: the 9.5 GB of rows tpr_constraint_params_batch writes,
    solved from those arrays (full iteration on every stage LP), against the fused default path (certified answers, rows
    never materialised) -- every K, sd^2, u and return code of every trajectory."""
    import torch
    dev = torch.device("cuda", 0)
    data = batch.make_synthetic_batch(65536, 7, 200)
    dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
    rows = batch.constraint_params_batch(*dv)
    got = batch.solve_dense_batch(rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], dv[2][1:] - dv[2][:-1])
    ref = batch.solve_batch(*dv)
    for k in ("K", "sd2", "u", "status"):
        assert torch.equal(torch.nan_to_num(got[k].double(), nan=-7.0), torch.nan_to_num(ref[k].double(), nan=-7.0)), k
    assert float((ref["status"] == 0).double().mean()) > 0.99


@pytest.mark.parametrize("B,d,N,interp,vel,acc", [(4096, 7, 200, True, True, True), (513, 7, 33, False, True, True), (64, 12, 50, True, True, True),
                                                  (40, 32, 9, True, True, True), (100, 3, 1, True, False, True), (77, 5, 20, True, True, False)])
def test_constraint_params_rows_are_the_wrappers(gpu, oracle, B, d, N, interp, vel, acc):
    """tpr_constraint_params_batch (one thread per element of a, b, c since round 4) against the restatement of
    seidelWrapper.__init__ (cy_seidel_solverwrapper.pyx:440-470): every element of a, b, c, low, high -- incl. the last
    gridpoint's repeated interpolation block, Collocation, no velocity / no acceleration constraint, non-uniform grids and
    knots, 32 dof, a single stage."""
    rng = np.random.default_rng(B + N)
    data = batch.make_synthetic_batch(B, d, N, seed=7 + d)
    grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)) * 0.98 + 0.01, [1.0]]) if N > 1 else data["grid"]
    vlim = data["vlim"] if vel else None
    alim = data["alim"] if acc else None
    rows = batch.constraint_params_batch(data["coef"], data["breaks"], grid, vlim, alim, interp)
    flags = (1 if vel else 0) | (2 if acc else 0) | (4 if interp else 0)
    for bsel in sorted(set([0, B - 1] + list(rng.integers(0, B, 6)))):
        w = oracle.Wrapper(data["coef"][bsel], data["breaks"], grid, None if vlim is None else vlim[bsel], None if alim is None else alim[bsel], flags=flags)
        assert rows["a"].shape[2] == w.nC
        wa, wb, wc = w.a_arr, w.b_arr, w.c_arr
        wa[:, :2] = 0; wb[:, :2] = 0; wc[:, :2] = 0   # rows 0, 1 (the x_next pair) are per-solve values: zeros in the entry's output
        assert_same(rows["a"][bsel], wa, "a"); assert_same(rows["b"][bsel], wb, "b"); assert_same(rows["c"][bsel], wc, "c")
        assert_same(rows["low"][bsel], w.low_arr, "low"); assert_same(rows["high"][bsel], w.high_arr, "high")

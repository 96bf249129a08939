"""The CPU oracle against outputs of the real reference (tests/golden, tools/make_golden.py):
bit-exact, including NaN patterns and return codes.  This pins the oracle on machines where the
reference itself is not importable (the GPU box)."""
import numpy as np
import pytest

from tests.helpers import assert_same, batch_fixtures, dense_constraints, dense_fixtures, fixture_problem, golden


@pytest.mark.parametrize("name", batch_fixtures())
def test_batch_fixture(oracle, name):
    fx = golden(name)
    coef, breaks, grid, vlim, alim, sd0, sd1, interp = fixture_problem(fx)
    flags = (oracle.FLAG_VEL if vlim is not None else 0) | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
    got = oracle.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, flags=flags)
    assert np.array_equal(got["status"], fx["status"])
    assert_same(got["K"], fx["K"], "K")
    assert_same(np.sqrt(got["sd2"]), fx["sd"], "sd")
    assert_same(got["u"], fx["u"], "u")
    if "X" in fx:
        for b in range(coef.shape[0]):
            w = oracle.Wrapper(coef[b], breaks, grid, None if vlim is None else vlim[b], alim[b], flags=flags)
            assert_same(w.compute_feasible_sets(), fx["X"][b], "X[%d]" % b)


@pytest.mark.parametrize("tag", ["n100", "auto"])
def test_example_kinematics(oracle, tag):
    fx = golden("example_kinematics_seed9")
    grid = fx[tag + "_grid"]
    w = oracle.Wrapper(fx["coef"][0], fx["breaks"], grid, fx["vlim"][0], fx["alim"][0])
    st, sdd, sd, xs, K = w.compute_parameterization(0.0, 0.0)
    assert st == int(fx[tag + "_status"]) == 0
    assert_same(K, fx[tag + "_K"], "K")
    assert_same(sd, fx[tag + "_sd"], "sd")
    assert_same(sdd, fx[tag + "_u"], "u")
    w2 = oracle.Wrapper(fx["coef"][0], fx["breaks"], grid, fx["vlim"][0], fx["alim"][0])
    assert_same(w2.compute_feasible_sets(), fx[tag + "_X"], "X")
    # constraint parameters: path derivatives, velocity bound, interpolated acceleration rows
    d = fx["coef"].shape[3]
    qs = np.array([oracle.path_eval(fx["coef"][0], fx["breaks"], s)[0] for s in grid])
    qss = np.array([oracle.path_eval(fx["coef"][0], fx["breaks"], s)[1] for s in grid])
    assert_same(qs, fx[tag + "_qs"], "qs")
    assert_same(qss, fx[tag + "_qss"], "qss")
    xb = np.array([oracle.velocity_xbound(q, fx["vlim"][0]) for q in qs])
    assert_same(xb, fx[tag + "_xbound"], "xbound")
    a, b = w.a_arr, w.b_arr
    assert_same(a[:, np.r_[2:2 + d, 2 + 2 * d:2 + 3 * d]], fx[tag + "_acc_a"], "a_intp")
    assert_same(b[:, np.r_[2:2 + d, 2 + 2 * d:2 + 3 * d]], fx[tag + "_acc_b"], "b_intp")
    # wrapper rows = F a, F b, F c - g
    F, g = fx[tag + "_acc_F"], fx[tag + "_acc_g"]
    assert_same(a[:, 2:], fx[tag + "_acc_a"] @ F.T, "F a")
    assert_same(w.c_arr[:, 2:], fx[tag + "_acc_c"] @ F.T - g, "F c - g")


@pytest.mark.parametrize("lp1d", [0, 1])
def test_stagewise_sequence(oracle, lp1d):
    """solve_stagewise_optim with the reference's stateful warm start, same call sequence."""
    fx = golden("example_kinematics_seed9")
    w = oracle.Wrapper(fx["coef"][0], fx["breaks"], fx["n100_grid"], fx["vlim"][0], fx["alim"][0],
                       solve_lp1d=lp1d)
    q, r = fx["stagewise_q_lp1d%d" % lp1d], fx["stagewise_r_lp1d%d" % lp1d]
    assert np.isnan(r).any() and np.isfinite(r).any()
    for row, want in zip(q, r):
        got = w.solve_stagewise_optim(int(row[0]), None, row[1:3], *row[3:7])
        assert_same(got, want, "stagewise")


def test_random_lps(oracle):
    fx = golden("random_lps")
    for t in range(fx["v"].shape[0]):
        res, val, var, ac = oracle.lp2d(fx["v"][t], fx["a"][t], fx["b"][t], fx["c"][t], fx["low"][t],
                                        fx["high"][t], fx["active_in"][t])
        assert res == fx["result"][t]
        if res:
            assert val == fx["optval"][t] and list(var) == list(fx["optvar"][t])
            assert list(ac) == list(fx["active_out"][t])
    for t in range(fx["v1"].shape[0]):
        res, val, var, ac = oracle.lp1d(fx["v1"][t], fx["a1"][t], fx["b1"][t], fx["low1"][t], fx["high1"][t])
        assert res == fx["result1"][t]
        if res:
            assert val == fx["optval1"][t] and var == fx["optvar1"][t] and ac == fx["active1"][t]


@pytest.mark.parametrize("name", ["sd_batch_d5_N80", "sd_batch_d10_N50"])
def test_sd_fixture_oracle(oracle, name):
    """TOPPRAsd restatement (oracle) vs the reference's outputs."""
    fx = golden(name)
    out = oracle.solve_batch_sd(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["desired"],
                                fx["sd_start"], fx["sd_end"])
    assert np.array_equal(out["status"], fx["status"])
    assert_same(out["K"], fx["K"], "K")
    assert_same(np.sqrt(out["sd2"]), fx["sd"], "sd")
    assert_same(out["u"], fx["u"], "u")


@pytest.mark.parametrize("name", ["reach_d5_N60", "reach_d3_N40_collocation"])
def test_oracle_reachable_sets_vs_reference_fixture(oracle, name):
    """compute_reachable_sets of the oracle against the real reference's outputs (tools/make_golden.py)."""
    fx = golden(name)
    interp = bool(int(fx["interpolation"]))
    flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
    for b in range(fx["coef"].shape[0]):
        w = oracle.Wrapper(fx["coef"][b], fx["breaks"], fx["grid"], fx["vlim"][b], fx["alim"][b], flags=flags)
        L, X = w.compute_reachable_sets(float(fx["sdmin"][b]), float(fx["sdmax"][b]))
        assert_same(L, fx["L"][b], "L[%d]" % b)
        assert_same(X, fx["X"][b], "X[%d]" % b)


def test_oracle_wrapper_state_across_passes(oracle):
    """The oracle's wrapper object carries active_c_up / active_c_down across passes like the reference's: the chain
    compute_parameterization -> compute_feasible_sets -> compute_controllable_sets -> compute_parameterization of
    tests/golden/reuse_d7_N100 (ONE reference instance per trajectory), bit for bit."""
    from oracle.oracle import FLAG_ACC, FLAG_INTERP, FLAG_VEL
    fx = golden("reuse_d7_N100")
    for b in range(fx["coef"].shape[0]):
        flags = FLAG_VEL | FLAG_ACC | (FLAG_INTERP if int(fx["scheme"][b]) == 1 else 0)
        w = oracle.Wrapper(fx["coef"][b], fx["breaks"], fx["grid"], fx["vlim"][b], fx["alim"][b], flags=flags)
        for tag in ("1", None, "3"):
            if tag is None:
                assert np.array_equal(w.compute_feasible_sets(), fx["X"][b], equal_nan=True), b
                assert np.array_equal(w.compute_controllable_sets(float(fx["sdmin"][b]), float(fx["sdmax"][b])), fx["K2"][b],
                                      equal_nan=True), b
                continue
            st, sdd, sd, xs, K = w.compute_parameterization(float(fx["sd_start"][b]), float(fx["sd_end"][b]))
            assert st == int(fx["status" + tag][b]), (b, tag)
            assert np.array_equal(K, fx["K" + tag][b], equal_nan=True), (b, tag)
            if st == 0:
                assert np.array_equal(sd, fx["sd" + tag][b]) and np.array_equal(sdd, fx["u" + tag][b]), (b, tag)


@pytest.mark.parametrize("name", dense_fixtures())
def test_dense_fixture(oracle, name):
    """Constraint lists beyond velocity + acceleration (JointTorqueConstraint, SecondOrderConstraint): the oracle's
    seidelWrapper on the dense rows of the fixture -- the reference's constraint objects flattened by
    toppra_amd.solverwrapper.dense_rows -- reproduces what the REFERENCE returned for those objects: K, sd, u, return
    codes, feasible sets, controllable sets, bit for bit.  This pins both the dense oracle and the row assembly."""
    fx = golden(name)
    got = oracle.solve_dense_batch(fx["a"], fx["b"], fx["c"], fx["low"], fx["high"], fx["deltas"], fx["sd_start"], fx["sd_end"],
                                   want_X=True)
    assert np.array_equal(got["status"], fx["status"])
    assert_same(got["K"], fx["K"], "K")
    ok = fx["status"] == 0
    assert_same(got["sd"][ok], fx["sd"][ok], "sd")
    assert_same(got["u"][ok], fx["u"][ok], "u")
    assert np.isnan(fx["sd"][fx["status"] == 1]).all()
    assert_same(got["X"], fx["X"], "X")
    for b in range(fx["a"].shape[0]):
        w = oracle.DenseWrapper(fx["a"][b], fx["b"][b], fx["c"][b], fx["low"][b], fx["high"][b], fx["deltas"])
        assert_same(w.compute_controllable_sets(float(fx["sdmin_c"]), float(fx["sdmax_c"])), fx["Kc"][b], "Kc[%d]" % b)
        for key, lo, hi in (("L", 0.0, 0.3), ("L_point", 0.1, 0.1)):  # reachable sets (a point start: the 1-variable path first)
            w = oracle.DenseWrapper(fx["a"][b], fx["b"][b], fx["c"][b], fx["low"][b], fx["high"][b], fx["deltas"])
            assert_same(w.compute_reachable_sets(lo, hi)[0], fx[key][b], "%s[%d]" % (key, b))


@pytest.mark.parametrize("name", dense_fixtures())
def test_dense_fixture_desired_duration(oracle, name):
    """TOPPRAsd of the REFERENCE on the torque / second-order constraint lists (unachievably short, in-range and
    unachievably long desired durations, uncontrollable starts): the oracle on the fixture's dense rows, bit for bit."""
    fx = golden(name)
    got = oracle.solve_dense_batch_sd(fx["a"], fx["b"], fx["c"], fx["low"], fx["high"], fx["deltas"], fx["sd_desired"],
                                      fx["sd_start"], fx["sd_end"])
    assert np.array_equal(got["status"], fx["sd_status"])
    assert_same(got["K"], fx["sd_K"], "K")
    assert_same(got["sd"], fx["sd_sd"], "sd")
    assert_same(got["u"], fx["sd_u"], "u")


def test_dense_reuse_fixture(oracle):
    """Passes chained on ONE reference instance with a torque constraint (parameterization -> feasible sets -> controllable
    sets -> reachable sets -> parameterization): the oracle's DenseWrapper driven in the same order -- the warm-start state
    carried from pass to pass changes bits in 7 of the 16 trajectories."""
    fx = golden("dense_reuse_d5_N60")
    assert int(fx["differs"]) >= 3
    for b in range(fx["a"].shape[0]):
        w = oracle.DenseWrapper(fx["a"][b], fx["b"][b], fx["c"][b], fx["low"][b], fx["high"][b], fx["deltas"])
        st, sdd, sd, xs, K = w.compute_parameterization(float(fx["sd_start"][b]), float(fx["sd_end"][b]))
        assert st == 0
        assert_same(K, fx["K"][b], "K"); assert_same(sd, fx["sd"][b], "sd"); assert_same(sdd, fx["u"][b], "u")
        assert_same(w.compute_feasible_sets(), fx["X"][b], "X after the parameterization")
        assert_same(w.compute_controllable_sets(0.05, 0.4), fx["Kc"][b], "controllable sets after that")
        assert_same(w.compute_reachable_sets(0.0, 0.3)[0], fx["L"][b], "reachable sets after that")
        st, sdd, sd, xs, K = w.compute_parameterization(float(fx["sd_start"][b]), float(fx["sd_end"][b]))
        assert_same(K, fx["K2"][b], "K, second parameterization"); assert_same(sd, fx["sd2nd"][b], "sd, second parameterization")
        assert_same(sdd, fx["u2nd"][b], "u, second parameterization")


def test_dense_rows_from_the_mirror_constraint_classes():
    """toppra_amd's own SecondOrderConstraint / JointTorqueConstraint (host numpy through the user's inverse dynamics, as in
    the reference) + dense_rows against the rows the REFERENCE's constraint objects gave (fixtures without a velocity /
    acceleration constraint need no GPU): the same bits."""
    import toppra_amd as ta
    from toppra_amd.solverwrapper import dense_rows
    fx = golden("dense_torque_only_d3_N30")
    for b in range(fx["a"].shape[0]):
        path = ta.SplineInterpolator(fx["knots"], fx["way"][b])
        rows = dense_rows(dense_constraints(fx, b, ta.constraint), path, fx["grid"])
        for k in ("a", "b", "c", "low", "high"):
            assert_same(rows[k], fx[k][b], "%s[%d]" % (k, b))
        assert_same(rows["deltas"], fx["deltas"], "deltas")
    # the second-order class with per-gridpoint F, g (not `identical`), Collocation and Interpolation, against the
    # identical-F torque class on the same model: the same rows
    rng = np.random.default_rng(0)
    d = 4
    path = ta.SplineInterpolator(np.linspace(0, 1, 5), rng.standard_normal((5, d)))
    grid = np.linspace(0, 1, 21)
    from tests.helpers import torque_model
    inv_dyn = torque_model(1 + rng.random(d), rng.standard_normal(d), rng.standard_normal(d))
    taulim = np.stack([-5 - rng.random(d), 5 + rng.random(d)], axis=1)
    fric = 0.1 * rng.random(d)
    for scheme in (0, 1):
        DT = ta.constraint.DiscretizationType(scheme)
        r1 = dense_rows([ta.constraint.JointTorqueConstraint(inv_dyn, taulim, fric, discretization_scheme=DT)], path, grid)
        r2 = dense_rows([ta.constraint.SecondOrderConstraint.joint_torque_constraint(inv_dyn, taulim, fric, discretization_scheme=DT)], path, grid)
        assert r1["nC"] == r2["nC"] == 2 + (4 if scheme else 2) * d
        for k in ("a", "b", "c"):
            np.testing.assert_allclose(r1[k], r2[k], rtol=0, atol=1e-13)

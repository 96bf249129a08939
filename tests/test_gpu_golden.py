"""HIP path against outputs of the REAL reference (tests/golden): bit-exact K / sd / u / X /
status, through the C-ABI, for both kernel families.  Stated tolerance of the north star is
1e-8 on sd^2; exact equality is asserted because the kernels replicate the reference's arithmetic.
Fixtures cover d = 1 .. 16 (tools/make_golden.py)."""
import ctypes as C

import numpy as np
import pytest

from tests import kat_vectors as kat
from tests.helpers import assert_same, batch_fixtures, fixture_problem, golden
from toppra_amd import _capi, batch

pytestmark = pytest.mark.gpu


def _variants(d, interp, has_vel=True):
    # 1 generic lane kernel; 2 rows across lanes (8 lanes up to 8 dof, 16 above); 3 certified lane kernel;
    # 4 one trajectory per wave (the latency kernel: every dof and constraint set)
    # Collocation: the interpolation blocks are disabled rows (families 2, 4) / null rows (family 3)
    # (family 3 is instantiated up to 8 dof, family 2 up to 16)
    return [1, 2, 3, 4] if d <= 15 else ([1, 2, 4] if d <= 16 else [1, 4])


@pytest.mark.parametrize("name", batch_fixtures())
def test_batch_fixture(gpu, name):
    fx = golden(name)
    coef, breaks, grid, vlim, alim, sd0, sd1, interp = fixture_problem(fx)
    for variant in _variants(coef.shape[3], interp):
        got = batch.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, interp, want_sd=True, variant=variant)
        assert np.array_equal(got["status"], fx["status"]), variant
        assert_same(got["K"], fx["K"], "K v%d" % variant)
        assert_same(got["sd"], fx["sd"], "sd v%d" % variant)
        assert_same(got["u"], fx["u"], "u v%d" % variant)
        ok = fx["status"] == 0
        assert_same(got["sd2"][ok], fx["sd"][ok] ** 2, "sd2", atol=1e-8)
    if "X" in fx:
        assert_same(batch.feasible_sets_batch(coef, breaks, grid, vlim, alim, interp), fx["X"], "X")
        # ... and through every kernel family that serves it (3: the certified lane kernel, fast and sound certificates)
        for kw in ([dict(variant=2)] if coef.shape[3] <= 16 else []) + [dict(variant=4)] + ([dict(variant=3)] if coef.shape[3] <= 15 and alim is not None else []) + ([dict(variant=3, sound=True)] if coef.shape[3] <= 15 and alim is not None else []):
            assert_same(batch.feasible_sets_batch(coef, breaks, grid, vlim, alim, interp, **kw), fx["X"], "X %s" % kw)
    # compute_controllable_sets(sd_end, sd_end) is the K of the parameterization
    K = batch.controllable_sets_batch(coef, breaks, grid, vlim, alim, sd1, sd1, interp)
    assert_same(K, fx["K"], "controllable sets")


@pytest.mark.parametrize("tag", ["n100", "auto"])
def test_example_kinematics(gpu, tag):
    fx = golden("example_kinematics_seed9")
    grid = fx[tag + "_grid"]
    d = fx["coef"].shape[3]
    for variant in (1, 2, 3, 4):
        got = batch.solve_batch(fx["coef"], fx["breaks"], grid, fx["vlim"], fx["alim"], want_sd=True,
                                variant=variant)
        assert got["status"][0] == 0
        assert_same(got["K"][0], fx[tag + "_K"], "K")
        assert_same(got["sd"][0], fx[tag + "_sd"], "sd")
        assert_same(got["u"][0], fx[tag + "_u"], "u")
    assert_same(batch.feasible_sets_batch(fx["coef"], fx["breaks"], grid, fx["vlim"], fx["alim"])[0],
                fx[tag + "_X"], "X")
    par = batch.constraint_params_batch(fx["coef"], fx["breaks"], grid, fx["vlim"], fx["alim"])
    assert_same(par["qs"][0], fx[tag + "_qs"], "qs")
    assert_same(par["qss"][0], fx[tag + "_qss"], "qss")
    assert_same(par["xbound"][0], fx[tag + "_xbound"], "xbound")
    pick = np.r_[2:2 + d, 2 + 2 * d:2 + 3 * d]
    assert_same(par["a"][0][:, pick], fx[tag + "_acc_a"], "a_intp")
    assert_same(par["b"][0][:, pick], fx[tag + "_acc_b"], "b_intp")
    assert_same(par["c"][0][:, 2:], fx[tag + "_acc_c"] @ fx[tag + "_acc_F"].T - fx[tag + "_acc_g"], "c")
    assert_same(par["high"][0][:, 1], np.minimum(fx[tag + "_xbound"][:, 1], 1e8), "high")


def test_cpp_scenario(gpu):
    fx = golden("cpp_scenario_collocation")
    got = batch.solve_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], interpolation=False,
                            want_sd=True)
    assert got["status"][0] == 0
    assert_same(got["K"][0], fx["K"], "K")
    assert_same(got["sd"][0], fx["sd"], "sd")
    np.testing.assert_allclose(got["K"][0][:, 1], kat.CPP_K_MAX, atol=1e-6)
    np.testing.assert_allclose(got["sd2"][0], kat.CPP_SD2, atol=1e-6)
    X = batch.feasible_sets_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], False)
    assert_same(X[0], fx["X"], "X")
    np.testing.assert_allclose(X[0][:, 1], kat.CPP_X_MAX, atol=1e-6)


def _lp1d(v, a, b, low, high):
    n, rows = a.shape
    res = np.zeros(n, np.int32); act = np.zeros(n, np.int32)
    val = np.zeros(n); var = np.zeros(n)
    args = [np.ascontiguousarray(x, dtype=np.float64) for x in (v, a, b, low, high)]
    _capi.check(_capi.load().tpr_lp1d_batch(n, rows, *[_capi.ptr(x) for x in args], _capi.ptr(res),
                                            _capi.ptr(val), _capi.ptr(var), _capi.ptr(act), None))
    return res, val, var, act


def _lp2d(v, a, b, c, low, high, ac):
    n, rows = a.shape
    res = np.zeros(n, np.int32); aco = np.zeros((n, 2), np.int32)
    val = np.zeros(n); var = np.zeros((n, 2))
    args = [np.ascontiguousarray(x, dtype=np.float64) for x in (v, a, b, c, low, high)]
    ac = np.ascontiguousarray(ac, dtype=np.int32)
    _capi.check(_capi.load().tpr_lp2d_batch(n, rows, *[_capi.ptr(x) for x in args], _capi.ptr(ac),
                                            _capi.ptr(res), _capi.ptr(val), _capi.ptr(var), _capi.ptr(aco), None))
    return res, val, var, aco


def test_lp_kats(gpu):
    for v, a, b, low, high, res, optval, optvar, active in kat.LP1D:
        a = np.asarray(a, float).reshape(1, -1); b = np.asarray(b, float).reshape(1, -1)
        r = _lp1d(np.array([v], float), a, b, np.array([low], float), np.array([high], float))
        assert r[0][0] == res
        if res:
            assert r[1][0] == optval and r[2][0] == optvar and r[3][0] == active
    for v, a, b, c, low, high, ac, res, optval, optvar, ac_exp in kat.LP2D:
        a, b, c = [np.asarray([] if x is None else x, float).reshape(1, -1) for x in (a, b, c)]
        r = _lp2d(np.array([v], float), a, b, c, np.array([low], float), np.array([high], float), np.array([ac]))
        assert r[0][0] == res
        if res:
            np.testing.assert_allclose(r[1][0], optval)
            np.testing.assert_allclose(r[2][0], optvar)
            assert set(r[3][0].tolist()) == set(ac_exp)


def test_random_lps_match_reference(gpu):
    fx = golden("random_lps")
    res, val, var, aco = _lp2d(fx["v"], fx["a"], fx["b"], fx["c"], fx["low"], fx["high"], fx["active_in"])
    assert np.array_equal(res, fx["result"])
    ok = res == 1
    assert ok.any() and (~ok).any()
    assert_same(val[ok], fx["optval"][ok], "optval")
    assert_same(var[ok], fx["optvar"][ok], "optvar")
    assert np.array_equal(aco[ok], fx["active_out"][ok])
    res, val, var, act = _lp1d(fx["v1"], fx["a1"], fx["b1"], fx["low1"], fx["high1"])
    assert np.array_equal(res, fx["result1"])
    ok = res == 1
    assert_same(val[ok], fx["optval1"][ok], "optval1")
    assert_same(var[ok], fx["optvar1"][ok], "optvar1")
    assert np.array_equal(act[ok], fx["active1"][ok])


@pytest.mark.parametrize("name", ["reach_d5_N60", "reach_d3_N40_collocation"])
def test_reachable_sets_fixture(gpu, oracle, name):
    """compute_reachable_sets against the real reference's outputs (tools/make_golden.py: non-uniform grid,
    sdmin == sdmax and sdmin < sdmax starts, infeasible starts) and against the oracle; the single-trajectory
    drop-in method too."""
    import toppra_amd as ta
    fx = golden(name)
    interp = bool(int(fx["interpolation"]))
    L, X = batch.reachable_sets_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"], fx["sdmin"], fx["sdmax"],
                                      interp, want_X=True)
    assert np.isnan(fx["L"]).any()
    assert_same(L, fx["L"], "L")
    assert_same(X, fx["X"], "X")
    flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
    for b in (0, 3):
        w = oracle.Wrapper(fx["coef"][b], fx["breaks"], fx["grid"], fx["vlim"][b], fx["alim"][b], flags=flags)
        assert_same(w.compute_reachable_sets(float(fx["sdmin"][b]), float(fx["sdmax"][b]))[0], L[b], "oracle L")

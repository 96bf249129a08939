"""HIP path vs the CPU oracle on identical seeded inputs (runs on the MI355X box).

Tolerance: the north star asks for |sd^2 - sd^2_ref| <= 1e-8; the kernels are written to be
bit-exact with the oracle (which is bit-exact with the reference), so the tests assert exact
equality first and report the max deviation when that fails.
"""
import numpy as np
import pytest

from toppra_amd import batch

pytestmark = pytest.mark.gpu

ATOL = 1e-8


def _compare(got, ref, exact=True):
    assert np.array_equal(got["status"], ref["status"])
    for key in ("K", "sd2", "u"):
        g, r = np.asarray(got[key]), np.asarray(ref[key])
        assert np.array_equal(np.isnan(g), np.isnan(r)), key
        dev = np.nanmax(np.abs(g - r)) if np.isfinite(r).any() else 0.0
        assert dev <= ATOL, (key, dev)
        if exact:
            assert np.array_equal(g, r, equal_nan=True), (key, "not bit-exact, max dev %g" % dev)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("B,d,N", [(256, 7, 200), (64, 6, 500), (130, 3, 50), (65, 1, 20), (97, 8, 64),
                                   (40, 2, 33), (33, 4, 70), (50, 5, 101)])
def test_solve_batch_matches_oracle(gpu, oracle, B, d, N, variant):
    data = batch.make_synthetic_batch(B, d, N, seed=1234 + d)
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"],
                            variant=variant)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    assert (ref["status"] == 0).mean() > 0.9
    _compare(got, ref)


@pytest.mark.parametrize("variant", [1, 2])
def test_nonzero_boundary_velocities(gpu, oracle, variant):
    data = batch.make_synthetic_batch(128, 7, 100, seed=7)
    rng = np.random.default_rng(5)
    sd0, sd1 = 0.5 * rng.random(128), 0.5 * rng.random(128)
    sd0[::7] = 5.0  # some uncontrollable starts
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1,
                            variant=variant)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1)
    assert set(np.unique(ref["status"])) >= {0, 1}
    _compare(got, ref)


def test_collocation_and_single_constraint(gpu, oracle):
    data = batch.make_synthetic_batch(64, 5, 80, seed=11)
    from oracle.oracle import FLAG_ACC, FLAG_VEL
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"],
                            interpolation=False)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"],
                             flags=FLAG_VEL | FLAG_ACC)
    _compare(got, ref)
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], None, data["alim"])
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], None, data["alim"],
                             flags=FLAG_ACC | 4)
    _compare(got, ref)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("B,d,N,nway", [(9, 2, 1, 2), (17, 3, 2, 3), (33, 7, 3, 2), (64, 1, 5, 4), (1, 7, 200, 5),
                                        (31, 8, 7, 9), (257, 6, 33, 6)])
def test_edge_shapes(gpu, oracle, B, d, N, nway, variant):
    """Smallest grids (N = 1), one spline segment, single trajectory, ragged batch sizes."""
    data = batch.make_synthetic_batch(B, d, N, seed=B + N, n_waypoints=nway)
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=variant)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    _compare(got, ref)


def test_empty_batch_and_bad_arguments(gpu):
    data = batch.make_synthetic_batch(4, 3, 10)
    out = batch.solve_batch(data["coef"][:0], data["breaks"], data["grid"], data["vlim"][:0], data["alim"][:0])
    assert out["sd2"].shape == (0, 11) and out["status"].shape == (0,)
    from toppra_amd import _capi
    with pytest.raises(_capi.ToppraHipError):  # d > TPR_MAX_DOF
        big = batch.make_synthetic_batch(2, 17, 10)
        batch.solve_batch(big["coef"], big["breaks"], big["grid"], big["vlim"], big["alim"])
    wide = batch.make_synthetic_batch(8, 12, 20)  # d > 8: 16 lanes per trajectory
    assert batch.solve_batch(wide["coef"], wide["breaks"], wide["grid"], wide["vlim"], wide["alim"],
                             variant=2)["status"].shape == (8,)
    with pytest.raises(_capi.ToppraHipError):  # the fast kernel refuses Collocation when forced
        batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"],
                          interpolation=False, variant=2)


@pytest.mark.parametrize("B,d,N,nway", [(40, 9, 50, 5), (33, 12, 64, 6), (20, 16, 40, 5), (64, 14, 100, 4),
                                        (24, 7, 120, 40), (16, 3, 300, 120), (12, 16, 60, 64)])
def test_wide_dof_and_long_splines(gpu, oracle, B, d, N, nway):
    """d = 9..16 run with 16 lanes per trajectory; paths with many spline segments keep the
    coefficient table in global memory instead of LDS.  Both against the oracle, both families."""
    data = batch.make_synthetic_batch(B, d, N, seed=d * 100 + nway, n_waypoints=nway)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    for variant in (1, 2):
        got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=variant)
        _compare(got, ref)
    strict = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], strict=True)
    _compare(strict, ref)

"""HIP path vs the CPU oracle on identical seeded inputs (runs on the MI355X box).

Tolerance: the north star asks for |sd^2 - sd^2_ref| <= 1e-8; the kernels are written to be
bit-exact with the oracle (which is bit-exact with the reference), so the tests assert exact
equality first and report the max deviation when that fails.
"""
import numpy as np
import pytest

from toppra_amd import batch
from tests.helpers import need_reference_solver

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


@pytest.mark.parametrize("variant", [1, 2, 4])
@pytest.mark.parametrize("B,d,N", [(256, 7, 200), (64, 6, 500), (130, 3, 50), (65, 1, 20), (97, 8, 64),
                                   (40, 2, 33), (33, 4, 70), (50, 5, 101)])
def test_solve_batch_matches_oracle(gpu, oracle, B, d, N, variant):
    data = batch.make_synthetic_batch(B, d, N, seed=1234 + d)
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"],
                            variant=variant)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    assert (ref["status"] == 0).mean() > 0.9
    _compare(got, ref)


@pytest.mark.parametrize("variant", [1, 2, 4])
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


@pytest.mark.parametrize("variant", [1, 2, 4])
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
        big = batch.make_synthetic_batch(2, 33, 10)
        batch.solve_batch(big["coef"], big["breaks"], big["grid"], big["vlim"], big["alim"])
    wide = batch.make_synthetic_batch(8, 12, 20)  # d > 8: 16 lanes per trajectory
    assert batch.solve_batch(wide["coef"], wide["breaks"], wide["grid"], wide["vlim"], wide["alim"],
                             variant=2)["status"].shape == (8,)
    with pytest.raises(_capi.ToppraHipError):  # the certified lane kernel needs an acceleration constraint
        batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], None, variant=3)
    with pytest.raises(_capi.ToppraHipError):  # rows across lanes stop at 16 dof when forced
        big = batch.make_synthetic_batch(2, 20, 10)
        batch.solve_batch(big["coef"], big["breaks"], big["grid"], big["vlim"], big["alim"], variant=2)


@pytest.mark.parametrize("B,d,N", [(200, 17, 40), (130, 24, 60), (70, 32, 30)])
def test_dof_17_to_32_on_the_generic_kernel(gpu, oracle, B, d, N):
    """The reference has no dof limit; 17..32 dof run on the generic lane-per-trajectory kernel and on the
    one-trajectory-per-wave kernel (two / three row slots per lane; auto for these batch sizes)."""
    data = batch.make_synthetic_batch(B, d, N, seed=d)
    rng = np.random.default_rng(d)
    sd1 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1)
    got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1)
    _compare(got, ref)
    for variant in (1, 4):
        _compare(batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1,
                                   variant=variant), ref)
    X = batch.feasible_sets_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    K = batch.controllable_sets_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd1, sd1)
    assert np.array_equal(K, ref["K"], equal_nan=True) and X.shape == K.shape


@pytest.mark.parametrize("B,d,N,nway", [(40, 9, 50, 5), (33, 12, 64, 6), (20, 16, 40, 5), (64, 14, 100, 4),
                                        (24, 7, 120, 40), (16, 3, 300, 120), (12, 16, 60, 64)])
def test_wide_dof_and_long_splines(gpu, oracle, B, d, N, nway):
    """d = 9..16 run with 16 lanes per trajectory; paths with many spline segments keep the
    coefficient table in global memory instead of LDS.  Both against the oracle, both families."""
    data = batch.make_synthetic_batch(B, d, N, seed=d * 100 + nway, n_waypoints=nway)
    ref = oracle.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    for variant in (1, 2, 4):
        got = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], variant=variant)
        _compare(got, ref)
    strict = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], strict=True)
    _compare(strict, ref)


@pytest.mark.parametrize("B,d,N", [(4096, 7, 200), (700, 3, 60), (300, 12, 40)])
@pytest.mark.parametrize("mode", ["collocation", "acc_only", "vel_only", "collocation_no_vel"])
def test_fast_kernels_serve_every_constraint_set(gpu, oracle, B, d, N, mode):
    """Collocation and single-constraint problems run on the rows-across-lanes kernels (the missing
    acceleration blocks are disabled rows): identical bits to the generic lane kernel, to the full
    iteration and to the oracle, and the auto selection no longer falls back to the generic kernel."""
    from oracle.oracle import FLAG_ACC, FLAG_INTERP, FLAG_VEL
    data = batch.make_synthetic_batch(B, d, N, seed=77 + d)
    rng = np.random.default_rng(d)
    sd1 = np.where(rng.random(B) < 0.3, 0.2 * rng.random(B), 0.0)
    vlim = None if mode in ("acc_only", "collocation_no_vel") else data["vlim"]
    alim = None if mode == "vel_only" else data["alim"]
    interp = mode == "acc_only"
    flags = (FLAG_VEL if vlim is not None else 0) | (FLAG_ACC if alim is not None else 0) | (FLAG_INTERP if interp else 0)
    args = (data["coef"], data["breaks"], data["grid"], vlim, alim, None, sd1, interp)
    lane = batch.solve_batch(*args, variant=1)
    kws = [dict(variant=2), dict(variant=2, strict=True), dict(), dict(variant=4), dict(variant=4, strict=True)]
    if alim is not None and d <= 15:
        kws.append(dict(variant=3))  # the certified lane kernel: Interpolation and Collocation, velocity optional
    for kw in kws:
        got = batch.solve_batch(*args, **kw)
        for k in ("K", "sd2", "u", "status"):
            assert np.array_equal(got[k], lane[k], equal_nan=True), (kw, k)
    idx = np.arange(0, B, max(1, B // 128))
    ref = oracle.solve_batch(data["coef"][idx], data["breaks"], data["grid"], None if vlim is None else vlim[idx],
                             None if alim is None else alim[idx], None, sd1[idx], flags=flags)
    _compare({k: lane[k][idx] for k in ("K", "sd2", "u", "status")}, ref)
    X = batch.feasible_sets_batch(data["coef"][:64], data["breaks"], data["grid"], None if vlim is None else vlim[:64],
                                  None if alim is None else alim[:64], interp)
    assert X.shape == (64, N + 1, 2) and not np.isnan(X).any()


@pytest.mark.parametrize("d", [9, 10, 11, 12, 13, 14, 15])
def test_certified_lane_kernel_above_8_dof(gpu, oracle, d):
    """Family 3 serves 9..13 dof too (slim blocks, internal row numbering with a block stride of 16, three mask words for the
    80 row numbers), with the same trace-following certificates as below -- there is no other mode: solve (scaled paths,
    boundary velocities, Collocation), feasible sets and TOPPRAsd against the rows-across-lanes kernels -- the full iteration
    where there is a strict mode -- bit for bit, explicitly and through the default choice."""
    B, N = 1200, 50
    data = batch.make_synthetic_batch(B, d, N, seed=60 + d)
    rng = np.random.default_rng(d)
    scale = np.where(rng.random((B, 1, 1, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1)))
    sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
    sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
    for interp in (True, False):
        args = (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1, interp)
        full = batch.solve_batch(*args, variant=2, strict=True)
        for kw in (dict(variant=3), dict(variant=3, sound=True), dict()):
            got = batch.solve_batch(*args, **kw)
            for k in ("K", "sd2", "u", "status"):
                assert np.array_equal(got[k], full[k], equal_nan=True), (k, interp, kw)
        # ... and the truth is not only another GPU kernel: the first 96 trajectories against the CPU restatement of the reference
        m = 96
        flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
        ref = oracle.solve_batch(args[0][:m], args[1], args[2], args[3][:m], args[4][:m], sd0[:m], sd1[:m], flags=flags, nthreads=0)
        assert np.array_equal(ref["status"], full["status"][:m]), interp
        for k in ("K", "sd2", "u"):
            assert np.array_equal(ref[k], full[k][:m], equal_nan=True), (k, interp)
        fargs = args[:5] + (interp,)
        assert np.array_equal(batch.feasible_sets_batch(*fargs, variant=3), batch.feasible_sets_batch(*fargs, variant=2, strict=True), equal_nan=True)
        desired = rng.uniform(0.3, 6.0, size=B)
        want = batch.solve_desired_duration_batch(*args[:5], desired, sd0, sd1, variant=2, interpolation=interp)
        got = batch.solve_desired_duration_batch(*args[:5], desired, sd0, sd1, variant=3, interpolation=interp)
        for k in ("K", "sd2", "u", "status", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, interp)


@pytest.mark.parametrize("B,d,N", [(1, 9, 7), (65, 13, 5), (3, 12, 2), (130, 10, 1)])
def test_slim_blocks_partial_and_tiny(gpu, oracle, B, d, N):
    """The slim blocks of family 3 above 8 dof read the acceleration limits (their own in a row fetch, another lane's in the
    cooperative batches) from global memory and flush K two stages at a time: partial blocks (idle lanes shadow the last
    trajectory), odd and tiny stage counts."""
    data = batch.make_synthetic_batch(B, d, N, seed=7 * d + N)
    rng = np.random.default_rng(B)
    sd1 = np.where(rng.random(B) < 0.5, 0.2 * rng.random(B), 0.0)
    scale = 10.0 ** rng.uniform(-4, 0, size=(B, 1, 1, 1))
    for interp in (True, False):
        args = (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1, interp)
        full = batch.solve_batch(*args, variant=2, strict=True)
        got = batch.solve_batch(*args, variant=3)
        for k in ("K", "sd2", "u", "status"):
            assert np.array_equal(got[k], full[k], equal_nan=True), (k, interp)
        flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if interp else 0)
        ref = oracle.solve_batch(args[0], args[1], args[2], args[3], args[4], None, sd1, flags=flags, nthreads=0)  # (the CPU restatement)
        assert np.array_equal(ref["status"], got["status"]), interp
        for k in ("K", "sd2", "u"):
            assert np.array_equal(ref[k], got[k], equal_nan=True), (k, interp)
        fargs = args[:5] + (interp,)
        assert np.array_equal(batch.feasible_sets_batch(*fargs, variant=3), batch.feasible_sets_batch(*fargs, variant=2, strict=True), equal_nan=True)
        K = batch.controllable_sets_batch(*args[:5], 0.0, sd1, interp, variant=3)
        assert np.array_equal(K, batch.controllable_sets_batch(*args[:5], 0.0, sd1, interp, variant=2, strict=True), equal_nan=True)


def test_violated_row_with_zero_normal_is_an_infeasible_stage(gpu, oracle):
    """A joint that stands still (q' = q'' = 0 along the whole path) with an acceleration range that excludes 0 gives rows
    (0, 0, c > 0): violated, and not a line.  The reference raises ZeroDivisionError there (cy_seidel_solverwrapper.pyx:290,
    cdivision=False); here the LP is reported infeasible -- the one deliberate difference -- by every
    kernel family and by the oracle alike, and the other trajectories of the batch are not affected."""
    B, d, N = 96, 3, 40
    data = batch.make_synthetic_batch(B, d, N, seed=77)
    way = data["waypoints"].copy()
    hit = np.zeros(B, bool)
    hit[[0, 5, 17, 63, 64, 95]] = True
    way[hit, :, 1] = way[hit, :1, 1]                       # joint 1 does not move
    coef, breaks = batch.spline_coefficients(data["knots"], way)
    assert np.all(coef[hit][:, :3, :, 1] == 0.0)           # its q', q'' are exactly 0
    alim = data["alim"].copy()
    alim[hit, 1, 0], alim[hit, 1, 1] = 0.5, 2.0            # 0 is not an admissible acceleration for it
    args = (coef, breaks, data["grid"], data["vlim"], alim)
    want = oracle.solve_batch(*args, None, None)
    assert (want["status"][hit] != 0).all() and (want["status"][~hit] == 0).all()
    for kw in (dict(variant=1), dict(variant=2), dict(variant=2, strict=True), dict(variant=3), dict(variant=4), dict()):
        got = batch.solve_batch(*args, **kw)
        for k in ("K", "sd2", "u", "status"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (kw, k)
        assert np.isnan(got["sd2"][hit]).all() and not np.isnan(got["sd2"][~hit]).any()


@pytest.mark.parametrize("d,N,seed", [(7, 120, 11), (3, 40, 12), (12, 50, 13), (6, 200, 14)])
def test_against_the_references_own_compiled_solver(gpu, d, N, seed):
    """Not the restatement: the reference's compiled cy_seidel_solverwrapper (oracle/_ref, built from the sources under
    /root/reference and shipped as a binary) driven by the reference's two passes (oracle/ref_solver_baseline.py), against
    every kernel family -- K, sd, u bit for bit, failures included; scaled paths and non-zero boundary velocities."""
    rb = need_reference_solver()
    B = 24
    data = batch.make_synthetic_batch(B, d, N, seed=seed)
    rng = np.random.default_rng(seed)
    coef = data["coef"] * np.where(rng.random((B, 1, 1, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-5, 0.3, size=(B, 1, 1, 1)))
    # (boundary velocities on a 2^-10 lattice: their squares are exact, so Python's ** in the reference's passes and the
    # device's sd * sd cannot differ in the last bit -- the pow quirk has its own fixture, pow_boundary_d3_N40)
    sd0 = np.where(rng.random(B) < 0.4, np.round(0.2 * rng.random(B) * 1024) / 1024, 0.0)
    sd1 = np.where(rng.random(B) < 0.4, np.round(0.2 * rng.random(B) * 1024) / 1024, 0.0)
    ref = []
    for k in range(B):
        vel, acc = rb.constraint_tuples(coef[k], data["breaks"], data["grid"], data["vlim"][k], data["alim"][k])
        w = rb.make_wrapper([rb.PrecomputedConstraint(vel, False), rb.PrecomputedConstraint(acc, True)], None, data["grid"])
        ref.append(rb.parameterization(w, float(sd0[k]), float(sd1[k])))
    kws = [dict(variant=1), dict(variant=2), dict(variant=4), dict()] + ([dict(variant=3)] if d <= 15 else [])
    for kw in kws:
        got = batch.solve_batch(coef, data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1, want_sd=True, **kw)
        for k in range(B):
            sdd, sd, K = ref[k]
            if sd is None:  # FailUncontrollable: no profile in the reference
                assert got["status"][k] != 0, (kw, k)
                continue
            assert np.array_equal(got["K"][k], K, equal_nan=True), (kw, k, "K")
            assert np.array_equal(got["sd"][k], sd, equal_nan=True), (kw, k, "sd")
            if not np.isnan(sd).any():
                assert np.array_equal(got["u"][k], sdd), (kw, k, "u")


@pytest.mark.parametrize("d,N,seed", [(7, 80, 21), (4, 30, 22), (11, 40, 23)])
def test_sets_against_the_references_own_compiled_solver(gpu, d, N, seed):
    """compute_feasible_sets and compute_controllable_sets(sdmin, sdmax) of the reference's compiled solver under the
    reference's loops (oracle/ref_solver_baseline.py) against the HIP entries, every family, bit for bit."""
    rb = need_reference_solver()
    B = 16
    data = batch.make_synthetic_batch(B, d, N, seed=seed)
    rng = np.random.default_rng(seed)
    coef = data["coef"] * np.where(rng.random((B, 1, 1, 1)) < 0.5, 1.0, 10.0 ** rng.uniform(-4, 0.3, size=(B, 1, 1, 1)))
    lo = np.round(0.1 * rng.random(B) * 1024) / 1024
    hi = lo + np.round(0.3 * rng.random(B) * 1024) / 1024
    Xr, Kr = [], []
    for k in range(B):
        vel, acc = rb.constraint_tuples(coef[k], data["breaks"], data["grid"], data["vlim"][k], data["alim"][k])
        cons = [rb.PrecomputedConstraint(vel, False), rb.PrecomputedConstraint(acc, True)]
        Xr.append(rb.feasible_sets(rb.make_wrapper(cons, None, data["grid"])))
        Kr.append(rb.controllable_sets(rb.make_wrapper(cons, None, data["grid"]), float(lo[k]), float(hi[k])))
    args = (coef, data["breaks"], data["grid"], data["vlim"], data["alim"])
    for kw in [dict(variant=1), dict(variant=2), dict(variant=4), dict()] + ([dict(variant=3)] if d <= 15 else []):
        X = batch.feasible_sets_batch(*args, True, **kw)
        K = batch.controllable_sets_batch(*args, lo, hi, True, **kw)
        for k in range(B):
            assert np.array_equal(X[k], Xr[k], equal_nan=True), (kw, k, "X")
            assert np.array_equal(K[k], Kr[k], equal_nan=True), (kw, k, "K")

"""TOPPRAsd (desired duration) on the GPU vs the reference's outputs (tests/golden) and the oracle."""
import numpy as np
import pytest

import toppra_amd as ta
from tests.helpers import assert_same, golden
from toppra_amd import batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["sd_batch_d5_N80", "sd_batch_d10_N50"])
def test_sd_fixture(gpu, name):
    fx = golden(name)
    out = batch.solve_desired_duration_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"],
                                             fx["desired"], fx["sd_start"], fx["sd_end"])
    assert np.array_equal(out["status"], fx["status"]) and set(fx["status"]) == {0, 1}
    assert_same(out["K"], fx["K"], "K")
    assert_same(out["sd"], fx["sd"], "sd")
    assert_same(out["u"], fx["u"], "u")
    ok = fx["status"] == 0
    assert np.any(out["alpha"][ok] == 1.0) and np.any((out["alpha"][ok] > 0) & (out["alpha"][ok] < 1))
    # ... and through each kernel family: rows across lanes for both scans (2), the fused certified lane launch (3, <= 13 dof)
    for variant in [2] + ([3] if fx["coef"].shape[3] <= 15 else []):
        got = batch.solve_desired_duration_batch(fx["coef"], fx["breaks"], fx["grid"], fx["vlim"], fx["alim"],
                                                 fx["desired"], fx["sd_start"], fx["sd_end"], variant=variant)
        for k in ("K", "sd", "u", "sd2", "alpha"):
            assert_same(got[k], out[k], "%s variant %d" % (k, variant))
        assert np.array_equal(got["status"], fx["status"])


def test_sd_fused_launch_matches_the_two_scan_path_and_the_oracle(gpu, oracle):
    """TOPPRAsd at batch size: the certified lane kernel runs the backward scan and both forward profiles in one launch
    (cert_solve_kernel<SDFWD>), the wave-per-trajectory kernel bisects and blends; every output incl. alpha must equal
    the rows-across-lanes path bit for bit (scaled paths, boundary velocities, desired durations on both sides of the
    reachable range), and a sample the oracle."""
    B, d, N = 20000, 7, 120
    data = batch.make_synthetic_batch(B, d, N, seed=31)
    rng = np.random.default_rng(31)
    scale = np.where(rng.random((B, 1, 1, 1)) < 0.6, 1.0, 10.0 ** rng.uniform(-5, 0, size=(B, 1, 1, 1)))
    sd0 = np.where(rng.random(B) < 0.3, 0.1 * rng.random(B), 0.0)
    sd1 = np.where(rng.random(B) < 0.3, 0.3 * rng.random(B), 0.0)
    desired = rng.uniform(0.3, 6.0, size=B)
    args = (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], desired, sd0, sd1)
    want = batch.solve_desired_duration_batch(*args, variant=2)
    for variant in (0, 3):   # (auto picks the fused launch at this size)
        got = batch.solve_desired_duration_batch(*args, variant=variant)
        for k in ("K", "sd2", "sd", "u", "alpha"):
            assert np.array_equal(got[k], want[k], equal_nan=True), (k, variant)
        assert np.array_equal(got["status"], want["status"])
    assert set(np.unique(want["status"])) >= {0, 1} and np.any((want["alpha"] > 0) & (want["alpha"] < 1))
    n = 24
    ref = oracle.solve_batch_sd(*[a[:n] if isinstance(a, np.ndarray) and a.shape[:1] == (B,) else a for a in args])
    for k in ("sd2", "u", "alpha"):
        assert np.array_equal(want[k][:n], ref[k], equal_nan=True), k
    assert np.array_equal(want["status"][:n], ref["status"])


def test_sd_long_grid_uses_the_thread_per_trajectory_finish(gpu, oracle):
    """N beyond the LDS rows of the wave-per-trajectory finish kernel (5 (N + 1) doubles <= 64 KB): decide / bisect /
    blend kernels, oracle parity."""
    B, d, N = 6, 3, 1700
    data = batch.make_synthetic_batch(B, d, N, seed=9)
    desired = np.array([0.5, 2.0, 3.0, 4.0, 6.0, 50.0])
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired)
    got = batch.solve_desired_duration_batch(*args)
    ref = oracle.solve_batch_sd(*args)
    assert np.array_equal(got["status"], ref["status"])
    for k in ("sd2", "u", "alpha"):
        assert np.array_equal(got[k], ref[k], equal_nan=True), k


def test_sd_class_and_oracle(gpu, oracle):
    data = batch.make_synthetic_batch(96, 7, 60, seed=12)
    rng = np.random.default_rng(2)
    fast = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], want_sd=True)
    ts, _ = batch.const_accel_times_batch(data["grid"], fast["sd"])
    desired = ts[:, -1] * rng.choice([0.7, 1.3, 2.5, 50.0], size=96)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], desired)
    got = batch.solve_desired_duration_batch(*args)
    ref = oracle.solve_batch_sd(*args)
    assert np.array_equal(got["status"], ref["status"])
    assert_same(got["sd2"], ref["sd2"], "sd2"); assert_same(got["u"], ref["u"], "u"); assert_same(got["alpha"], ref["alpha"], "alpha")
    # achieved durations hit the target within the bisection tolerance where it is reachable
    ts2, _ = batch.const_accel_times_batch(data["grid"], got["sd"])
    mid = (got["alpha"] > 0) & (got["alpha"] < 1)
    assert mid.any() and np.all(np.abs(ts2[mid, -1] - desired[mid]) < 1e-3)
    # drop-in class
    b = 3
    path = ta.SplineInterpolator(data["knots"], data["waypoints"][b])
    inst = ta.algorithm.TOPPRAsd([ta.constraint.JointVelocityConstraint(data["vlim"][b]),
                                  ta.constraint.JointAccelerationConstraint(data["alim"][b])], path,
                                 gridpoints=data["grid"])
    inst.set_desired_duration(float(desired[b]))
    sdd, sd, _ = inst.compute_parameterization(0, 0)
    assert_same(sd, got["sd"][b], "sd_vec"); assert_same(sdd, got["u"][b], "sdd_vec")

"""GPU const-accel parametrizer (tpr_const_accel_*_batch) vs the reference's ParametrizeConstAccel
outputs stored in tests/golden, and vs the host mirror on a random batch."""
import numpy as np
import pytest

import toppra_amd as ta
from tests.helpers import golden
from toppra_amd import batch

pytestmark = pytest.mark.gpu


def test_const_accel_matches_reference(gpu):
    fx = golden("example_kinematics_seed9")
    sd = fx["n100_sd"][None]
    ts, us = batch.const_accel_times_batch(fx["n100_grid"], sd)
    assert np.array_equal(ts[0], fx["ca_ts"]) and np.array_equal(us[0], fx["ca_us"])
    for order in (0, 1, 2):
        q = batch.const_accel_eval_batch(fx["coef"], fx["breaks"], fx["n100_grid"], sd, ts, us,
                                         fx["ca_times"][None], order)
        np.testing.assert_allclose(q[0], fx["ca_q%d" % order], rtol=1e-12, atol=1e-12)


def test_const_accel_batch_vs_host_mirror(gpu):
    data = batch.make_synthetic_batch(64, 5, 80, seed=21)
    out = batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], want_sd=True)
    ts, us = batch.const_accel_times_batch(data["grid"], out["sd"])
    rng = np.random.default_rng(0)
    times = rng.random((64, 50)) * ts[:, -1:]
    q = [batch.const_accel_eval_batch(data["coef"], data["breaks"], data["grid"], out["sd"], ts, us, times, o)
         for o in (0, 1, 2)]
    for b in (0, 17, 63):
        path = ta.SplineInterpolator(data["knots"], data["waypoints"][b])
        ca = ta.ParametrizeConstAccel(path, data["grid"], out["sd"][b])
        assert np.array_equal(ca._ts, ts[b]) and np.array_equal(ca._us, us[b])
        for o in (0, 1, 2):
            np.testing.assert_allclose(q[o][b], ca(times[b], o), rtol=1e-11, atol=1e-11)

"""Kernel family 5 (two trajectories per wave, 32 lanes each: csrc/tpr_pair.hip.inc) -- the latency kernel for a few thousand
trajectories (BASELINE config 2).  Bit for bit against the reference-generated fixtures, the CPU restatement of the reference,
the full Seidel iteration of family 2 and family 4, on odd and even batches, every constraint set, strict mode, per-trajectory
grids, and the sliver family on which the reference itself fails (cy_seidel_solverwrapper.pyx:127,:342,:353-355)."""
import os
import sys

import numpy as np
import pytest

from tests.helpers import batch_fixtures, fixture_problem, golden
from toppra_amd import batch

pytestmark = pytest.mark.gpu
KEYS = ("K", "sd2", "u", "status")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same(got, want, what):
    for k in KEYS:
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k]), equal_nan=True), what + (k,)


@pytest.mark.parametrize("name", [n for n in batch_fixtures() if golden(n)["coef"].shape[3] <= 7])  # (family 5 serves 1..7 dof)
def test_reference_fixtures(gpu, name):
    fx = golden(name)
    for strict in (False, True):
        got = batch.solve_batch(*fixture_problem(fx), want_sd=True, variant=5, strict=strict)
        assert np.array_equal(got["K"], fx["K"], equal_nan=True), (name, strict, "K")
        assert np.array_equal(got["sd"], fx["sd"], equal_nan=True), (name, strict, "sd")
        assert np.array_equal(got["u"], fx["u"], equal_nan=True), (name, strict, "u")
        assert np.array_equal(got["status"], fx["status"]), (name, strict, "status")


def test_example_kinematics(gpu):
    fx = golden("example_kinematics_seed9")
    for tag in ("n100", "auto"):
        got = batch.solve_batch(fx["coef"], fx["breaks"], fx[tag + "_grid"], fx["vlim"], fx["alim"], want_sd=True, variant=5)
        assert got["status"][0] == 0
        assert np.array_equal(got["K"][0], fx[tag + "_K"]) and np.array_equal(got["sd"][0], fx[tag + "_sd"]) and np.array_equal(got["u"][0], fx[tag + "_u"])


@pytest.mark.parametrize("B,d,N", [(301, 7, 120), (257, 1, 40), (200, 2, 33), (255, 4, 70), (256, 6, 101), (1, 7, 100), (3, 7, 1), (5, 2, 2)])
def test_against_the_oracle_the_full_iteration_and_family_4(gpu, oracle, B, d, N):
    data = batch.make_synthetic_batch(B, d, N, seed=900 + d + N)
    rng = np.random.default_rng(d * 7 + N)
    sd0 = np.where(rng.random(B) < 0.3, np.round(0.1 * rng.random(B) * 1024) / 1024, 0.0)
    sd1 = np.where(rng.random(B) < 0.3, np.round(0.3 * rng.random(B) * 1024) / 1024, 0.0)
    scale = 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
    tight = data["alim"] * np.where(rng.random((B, 1, 1)) < 0.3, 0.02, 1.0)  # nearly uncontrollable: failures, retries
    cases = {"boundary": (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], sd0, sd1, True),
             "fast_start": (data["coef"], data["breaks"], data["grid"], data["vlim"], tight, 40.0 * sd0, sd1, True),
             "scaled": (data["coef"] * scale, data["breaks"], data["grid"], data["vlim"], data["alim"], None, None, True),
             "collocation": (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"], None, sd1, False),
             "acc_only": (data["coef"], data["breaks"], data["grid"], None, data["alim"], None, None, True),
             "vel_only": (data["coef"], data["breaks"], data["grid"], data["vlim"], None, None, None, False)}
    for name, args in cases.items():
        got = batch.solve_batch(*args, variant=5)
        same(got, batch.solve_batch(*args, variant=2, strict=True), (B, d, N, name, "full iteration"))
        same(got, batch.solve_batch(*args, variant=4), (B, d, N, name, "family 4"))
        same(batch.solve_batch(*args, variant=5, strict=True), got, (B, d, N, name, "strict"))
        if name in ("boundary", "fast_start", "scaled", "collocation"):
            flags = oracle.DEFAULT_FLAGS if args[7] else oracle.DEFAULT_FLAGS & ~oracle.FLAG_INTERP
            ref = oracle.solve_batch(args[0], args[1], args[2], args[3], args[4], args[5], args[6], flags=flags, nthreads=4)
            same(got, ref, (B, d, N, name, "oracle"))
        without_K = batch.solve_batch(*args, variant=5, want_sd=True, want_K=False)
        assert np.array_equal(without_K["sd2"], got["sd2"], equal_nan=True) and np.array_equal(without_K["sd"], np.sqrt(got["sd2"]), equal_nan=True)


def test_per_trajectory_grids_and_long_tables(gpu):
    B, d, N = 201, 7, 90
    data = batch.make_synthetic_batch(B, d, N, seed=5)
    rng = np.random.default_rng(5)
    grid_b = np.sort(np.concatenate([np.zeros((B, 1)), rng.random((B, N - 1)), np.ones((B, 1))], axis=1), axis=1)
    grid_b[:, 1:-1] = 0.5 * grid_b[:, 1:-1] + 0.5 * data["grid"][None, 1:-1]
    args = (data["coef"], np.repeat(data["breaks"][None], B, axis=0), grid_b, data["vlim"], data["alim"])
    same(batch.solve_batch(*args, variant=5), batch.solve_batch(*args, variant=2, strict=True), ("per-trajectory grids",))
    for B, d, N, nway in ((24, 7, 120, 40), (17, 3, 300, 120), (9, 7, 700, 5), (6, 5, 64, 200), (4, 6, 40, 400)):
        data = batch.make_synthetic_batch(B, d, N, seed=d * 100 + nway, n_waypoints=nway)
        args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
        same(batch.solve_batch(*args, variant=5), batch.solve_batch(*args, variant=2, strict=True), ("long", B, d, N, nway))


@pytest.mark.parametrize("B,d,N,seed", [(768, 7, 60, 101), (769, 4, 50, 102)])
def test_sliver_family(gpu, oracle, B, d, N, seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_sliver_hunt as sh
    (coef, breaks, grid, vlim, alim, sd0, sd1), _ = sh.family(B, d, N, seed)
    got = batch.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, variant=5)
    same(got, batch.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, variant=2, strict=True), ("sliver", "full iteration"))
    cut = lambda v: None if v is None else np.broadcast_to(np.asarray(v, dtype=float), (B,))[:256]
    ref = oracle.solve_batch(coef[:256], breaks, grid, vlim[:256], alim[:256], cut(sd0), cut(sd1), nthreads=4)
    same({k: np.asarray(got[k])[:256] for k in KEYS}, ref, ("sliver", "oracle"))


def test_config_2_and_the_automatic_choice(gpu):
    """BASELINE config 2 (4096 x 7 x 200) is family 5's by default; the headline batch is not."""
    data = batch.make_synthetic_batch(4096, 7, 200, seed=20240924)
    args = (data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    auto = batch.solve_batch(*args)
    same(auto, batch.solve_batch(*args, variant=5), ("C2", "variant 5"))
    same(auto, batch.solve_batch(*args, variant=2), ("C2", "family 2"))
    assert (auto["status"] == 0).all()
    with pytest.raises(Exception):
        big = batch.make_synthetic_batch(8, 9, 20)
        batch.solve_batch(big["coef"], big["breaks"], big["grid"], big["vlim"], big["alim"], variant=5)

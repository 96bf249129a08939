"""EVERY instantiation of the certified lane kernels (kernel family 3) is executed at least once against another kernel family.

Why: these kernels sit at the edge of the register file, and twice a value-identical change of unrelated
code made ONE instantiation return wrong results on every input (round 3: feasible sets at 11 / 13 dof; round 5: the forward
profiles of <5 dof, TOPPRAsd>) while its neighbours stayed right.  Such a defect is input-independent, so a small random batch
through each (dof, sd output, grid in LDS or per trajectory, Interpolation or Collocation) instantiation of cert_solve_kernel,
cert_feasible_kernel and the TOPPRAsd launch -- compared bit for bit with the rows-across-lanes kernels, which share no device code
with them beyond the row generation -- is a cheap net under all of them (15 dofs x 8 + 15 x 4 + 15 x 4 launches of 96 trajectories).
Round 6: the net is self-standing -- per dof one 96 x 48 batch of each discretisation also goes through the CPU restatement of
the reference (oracle/), so a defect in what the two kernel families SHARE (row generation, the division / square-root
sequences of tpr_device.hpp) cannot pass it either.  What a failure here usually is: profiles/r06_miscompile_root_cause.md."""
import numpy as np
import pytest

from toppra_amd import batch

pytestmark = pytest.mark.gpu
B, N = 96, 48  # one full 64-lane block and a partial one


def _problem(d, seed, per_traj_grid):
    data = batch.make_synthetic_batch(B, d, N, seed=seed)
    rng = np.random.default_rng(seed)
    grid = data["grid"]
    if per_traj_grid:  # TPR_GRID_PER_TRAJ: the grid is read from global memory instead of the block's LDS copy
        inner = np.sort(rng.random((B, N - 1)), axis=1) * 0.8 + 0.1
        grid = 0.5 * np.concatenate([np.zeros((B, 1)), inner, np.ones((B, 1))], axis=1) + 0.5 * np.linspace(0, 1, N + 1)[None, :]
    sd0 = np.where(rng.random(B) < 0.3, np.round(0.1 * rng.random(B) * 1024) / 1024, 0.0)
    sd1 = np.where(rng.random(B) < 0.3, np.round(0.2 * rng.random(B) * 1024) / 1024, 0.0)
    return data, grid, sd0, sd1


@pytest.mark.parametrize("d", range(1, 16))
def test_every_solve_instantiation(gpu, oracle, d):
    for per_traj_grid in (False, True):
        data, grid, sd0, sd1 = _problem(d, 700 + d, per_traj_grid)
        for interp in (True, False):
            for want_sd in (False, True):
                args = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"], sd0, sd1, interp)
                want = batch.solve_batch(*args, want_sd=want_sd, variant=2)
                got = batch.solve_batch(*args, want_sd=want_sd, variant=3)
                what = (d, per_traj_grid, interp, want_sd)
                assert np.array_equal(got["status"], want["status"]), what
                for k in ("K", "sd2", "u") + (("sd",) if want_sd else ()):
                    assert np.array_equal(got[k], want[k], equal_nan=True), what + (k,)
                assert (want["status"] == 0).mean() > 0.5, what
            if not per_traj_grid:  # ... and the rows-across-lanes answer against the CPU restatement of the reference
                ref = oracle.solve_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], sd0, sd1,
                                         flags=oracle.DEFAULT_FLAGS if interp else oracle.DEFAULT_FLAGS & ~oracle.FLAG_INTERP, nthreads=4)
                assert np.array_equal(ref["status"], want["status"]), (d, interp, "oracle status")
                for k in ("K", "sd2", "u"):
                    assert np.array_equal(ref[k], want[k], equal_nan=True), (d, interp, "oracle", k)


@pytest.mark.parametrize("d", range(1, 16))
def test_every_feasible_sets_and_toppra_sd_instantiation(gpu, d):
    for per_traj_grid in (False, True):
        data, grid, sd0, sd1 = _problem(d, 800 + d, per_traj_grid)
        desired = np.random.default_rng(d).uniform(0.5, 5.0, size=B)
        for interp in (True, False):
            what = (d, per_traj_grid, interp)
            X2 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=2)
            X3 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=3)
            assert np.array_equal(X3, X2, equal_nan=True), what + ("X",)
            a = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1,
                                                   variant=2, interpolation=interp)
            b = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, sd0, sd1,
                                                   variant=3, interpolation=interp)
            assert np.array_equal(a["status"], b["status"]), what
            for k in ("K", "sd2", "sd", "u", "alpha"):
                assert np.array_equal(a[k], b[k], equal_nan=True), what + (k,)


def _tight_joint_problem(d, seed):
    """4 d trajectories in which ONE joint's rows bind the stage LPs: joint j % d gets acceleration limits a fiftieth of the others'
    -- both signs, the upper one only (the + rows: c = -amax), the lower one only (the - rows: c = amin), or both with a velocity
    limit that binds as well.  Round 6's 15-dof incident (profiles/r06_dof15_unsplit_unit_incident.log) was a register tuple of ONE
    joint's limits restored with a stale low dword: wrong only where that joint's + row bound the forward LP, i.e. on a fifth of a
    random batch at isolated stages.  Here every joint's every kind of row binds at most stages of some trajectory."""
    Bt = 4 * d
    data = batch.make_synthetic_batch(Bt, d, N, seed=seed)
    alim = np.array(data["alim"], dtype=np.float64)   # [B][d][2] = (amin, amax)
    vlim = np.array(data["vlim"], dtype=np.float64)
    for j in range(Bt):
        k, kind = j % d, j // d
        if kind in (0, 1, 3):
            alim[j, k, 1] *= 0.02
        if kind in (0, 2, 3):
            alim[j, k, 0] *= 0.02
        if kind == 3:
            vlim[j, k] *= 0.05
    return dict(data, alim=alim, vlim=vlim)


@pytest.mark.parametrize("d", range(1, 16))
def test_every_joints_rows_bind_somewhere(gpu, oracle, d):
    data = _tight_joint_problem(d, 900 + d)
    desired = np.random.default_rng(50 + d).uniform(2.0, 40.0, size=4 * d)
    rng = np.random.default_rng(60 + d)
    inner = np.sort(rng.random((4 * d, N - 1)), axis=1) * 0.8 + 0.1
    own_grids = 0.5 * np.concatenate([np.zeros((4 * d, 1)), inner, np.ones((4 * d, 1))], axis=1) + 0.5 * np.linspace(0, 1, N + 1)[None, :]
    # every instantiation: grid shared (in LDS up to 8 dof) / per trajectory, Interpolation / Collocation, with / without the sd output
    for grid, interp in ((data["grid"], True), (data["grid"], False), (own_grids, True), (own_grids, False)):
        args = (data["coef"], data["breaks"], grid, data["vlim"], data["alim"], None, None, interp)
        want = batch.solve_batch(*args, variant=2)
        got = batch.solve_batch(*args, variant=3)
        got_sd = batch.solve_batch(*args, want_sd=True, variant=3)
        for k in ("status", "K", "sd2", "u"):
            assert np.array_equal(got_sd[k], want[k], equal_nan=True), (d, interp, grid.ndim, "family 3 with sd output", k)
        ref = oracle.solve_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], None, None,
                                 flags=oracle.DEFAULT_FLAGS if interp else oracle.DEFAULT_FLAGS & ~oracle.FLAG_INTERP, nthreads=4)
        assert (ref["status"] == 0).mean() > 0.5, (d, interp, grid.ndim)
        for k in ("status", "K", "sd2", "u"):
            assert np.array_equal(want[k], ref[k], equal_nan=True), (d, interp, "family 2 vs oracle", k)
            assert np.array_equal(got[k], ref[k], equal_nan=True), (d, interp, "family 3 vs oracle", k)
        for v in (1, 4) + ((5,) if d <= 7 else ()):  # (the other families on the same batch: one trajectory per lane / wave, two per wave)
            other = batch.solve_batch(*args, variant=v)
            for k in ("status", "K", "sd2", "u"):
                assert np.array_equal(other[k], ref[k], equal_nan=True), (d, interp, "family %d vs oracle" % v, k)
        X2 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=2)
        X3 = batch.feasible_sets_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], interp, variant=3)
        assert np.array_equal(X3, X2, equal_nan=True), (d, interp, "X")
        a = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, None, None,
                                               variant=2, interpolation=interp)
        b = batch.solve_desired_duration_batch(data["coef"], data["breaks"], grid, data["vlim"], data["alim"], desired, None, None,
                                               variant=3, interpolation=interp)
        for k in ("status", "K", "sd2", "sd", "u", "alpha"):
            assert np.array_equal(a[k], b[k], equal_nan=True), (d, interp, "TOPPRAsd", k)

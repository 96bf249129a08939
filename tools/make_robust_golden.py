"""Generate golden vectors for the ROBUST (conic) path -- BASELINE config 4 -- with the reference's own ECOS wrapper.

Runs only where BOTH are importable: the reference package (``/root/reference`` or ``$TOPPRA_REFERENCE``, through
oracle/ref_loader.py) and PyPI ``ecos`` (the reference's solver for these stage problems,
toppra/solverwrapper/ecos_solverwrapper.py:90-207).  ``ecos`` is NOT installed in the build container of this repository
and there is no network, so the script cannot run there: it exists so that the fixtures can be produced on any machine
that has both (``pip install ecos`` next to a checkout of hungpham2511/toppra v0.6.2) and the robust row of the parity
table can be pinned:

    TOPPRA_REFERENCE=/path/to/toppra python tools/make_robust_golden.py            # writes tests/golden/robust_ecos_*.npz
    python -m pytest tests/test_gpu_robust.py -m gpu -k ecos_fixture                 # on the MI355X box

What is stored per fixture: the inputs (spline coefficients, breakpoints, grid, limits, ellipsoid axes, discretisation,
boundary velocities) and the reference's outputs through ``TOPPRA(..., solver_wrapper="ecos")`` -- feasible sets X,
controllable sets K, the parameterization (sdd, sd) and the return code -- for every trajectory of a small batch drawn from
the benchmark's C4 generator (toppra_amd.batch.make_synthetic_batch, examples/plot_robust_kinematics.py:40-62).
ECOS is an interior-point method with feastol = abstol = reltol ~ 1e-8; tests/test_gpu_robust.py compares at 1e-6 on
K, X, sd^2 and 1e-4 (relative) on u.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [
    # name, B, dof, N, interpolation scheme of the robust constraint, ellipsoid axes, sd_end range, velocity constraint
    ("robust_ecos_d7_N100", 24, 7, 100, 1, [1e-3, 5e-2, 9e-3], 0.0, True),     # BASELINE config 4's shape
    ("robust_ecos_d3_N40_collocation", 16, 3, 40, 0, [1e-3, 5e-2, 9e-3], 0.3, True),
    ("robust_ecos_d6_N60_wide", 16, 6, 60, 1, [1e-2, 1e-1, 5e-2], 0.2, True),
    ("robust_ecos_d4_N60_no_velocity", 12, 4, 60, 1, [1e-3, 5e-2, 9e-3], 0.0, False),
]


def main():
    try:
        import ecos  # noqa: F401
    except ImportError:
        raise SystemExit("PyPI `ecos` is not importable here: the fixtures must be generated on a machine that has it "
                         "(the reference's robust path has no other solver: ecos_solverwrapper.py:192)")
    from oracle import ref_loader
    toppra = ref_loader.load()
    if toppra is None:
        raise SystemExit("the reference package is not importable (set TOPPRA_REFERENCE to a checkout of hungpham2511/toppra)")
    from toppra_amd import batch
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, B, d, N, scheme, ell, sd_end_max, has_vel in CASES:
        data = batch.make_synthetic_batch(B, d, N, seed=7000 + d)
        rng = np.random.default_rng(d)
        sd_end = sd_end_max * rng.random(B)
        grid = data["grid"]
        X = np.full((B, N + 1, 2), np.nan)
        K = np.full((B, N + 1, 2), np.nan)
        sd = np.full((B, N + 1), np.nan)
        sdd = np.full((B, N), np.nan)
        code = np.zeros(B, dtype=np.int32)
        for b in range(B):
            path = toppra.SplineInterpolator(data["knots"], data["waypoints"][b])
            acc = toppra.constraint.JointAccelerationConstraint(
                data["alim"][b], discretization_scheme=toppra.constraint.DiscretizationType(scheme))
            rob = toppra.constraint.RobustLinearConstraint(acc, ell, scheme)
            cons = ([toppra.constraint.JointVelocityConstraint(data["vlim"][b])] if has_vel else []) + [rob]
            inst = toppra.algorithm.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="ecos")
            X[b] = inst.compute_feasible_sets()
            K[b] = inst.compute_controllable_sets(sd_end[b], sd_end[b])
            res = inst.compute_parameterization(0.0, sd_end[b])
            code[b] = {"Ok": 0, "FailUncontrollable": 1}.get(inst.problem_data.return_code.name, 2)  # tpr status codes
            if res[0] is not None:
                sdd[b], sd[b] = res[0], res[1]
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), coef=data["coef"], breaks=data["breaks"], grid=grid,
                            vlim=data["vlim"] if has_vel else np.zeros((0,)), alim=data["alim"], ell=np.asarray(ell, dtype=float),
                            interpolation=np.int32(scheme), sd_end=sd_end, X=X, K=K, sd=sd, sdd=sdd, return_code=code,
                            ecos_version=str(getattr(__import__("ecos"), "__version__", "?")))
        print("%s: %d trajectories, %d Ok" % (name, B, int((code == 0).sum())))


if __name__ == "__main__":
    main()

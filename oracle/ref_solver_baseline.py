"""The REFERENCE'S OWN compiled seidel solver as a CPU baseline that can run where /root/reference is absent.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/): nothing under toppra_amd/ imports this.

``oracle/build_ref.py`` compiles the reference's two Cython modules from the sources where they lie under
/root/reference into ``oracle/_ref/`` (binaries only; git-ignored, shipped to the GPU box with the snapshot).  The solver
module, ``toppra.solverwrapper.cy_seidel_solverwrapper``, needs exactly one thing from the reference's Python package at
import time -- ``from ..constraint import ConstraintType`` (cy_seidel_solverwrapper.pyx:11) -- and at construction it talks
to constraint objects through ``get_constraint_type()``, ``compute_constraint_params(path, gridpoints)`` and
``identical`` (:437-455).  ``PrecomputedConstraint`` below answers those from stored tuples, so on a box without the
reference's sources the compiled solver is loaded under stand-in ``toppra`` / ``toppra.constraint`` /
``toppra.solverwrapper`` module objects created here (no file of the reference is read or copied).

What is timed is the reference's solver doing the reference's work: per trajectory one ``seidelWrapper`` (solve_lp1d=True,
reachability_algorithm.py:121-125) and the two passes of ``TOPPRA.compute_parameterization`` -- the backward
controllable-set scan (reachability_algorithm.py:166-238: two ``solve_stagewise_optim`` calls per stage with objectives
(1e-9, -1) and (-1e-9, 1)) and the forward scan (:303-363 with time_optimal_algorithm.py:55-92: one call per stage with
objective (-2 delta_i, -1), the retry / shrink / clamp rules) -- restated below as plain Python loops, which is what they
are in the reference (3 N Python -> Cython crossings per trajectory).
"""
import enum
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np

from . import build_ref

TINY, SMALL, MAX_TRIES = 1e-8, 1e-5, 10  # toppra/constants.py:14-27 (the values the passes below use)
_solver = None


class _StandInConstraintType(enum.Enum):
    """All the compiled solver does with ConstraintType is compare a constraint's answer with .CanonicalLinear (:438)."""
    Unknown = -1
    CanonicalLinear = 0
    CanonicalConic = 1


def available() -> bool:
    return os.path.exists(build_ref.so_path("toppra.solverwrapper.cy_seidel_solverwrapper"))


def load():
    """The compiled ``cy_seidel_solverwrapper`` module.  With /root/reference present the real package is used
    (oracle/ref_loader.py); without it, stand-in parent modules carrying the mirror's ConstraintType."""
    global _solver
    if _solver is not None:
        return _solver
    name = "toppra.solverwrapper.cy_seidel_solverwrapper"
    if build_ref.have_reference() and not os.environ.get("TPR_REF_FORCE_STANDIN"):
        from . import ref_loader
        if ref_loader.load() is not None:
            _solver = importlib.import_module(name)
            return _solver
    if not available():
        raise RuntimeError("oracle/_ref holds no compiled reference solver (build it where /root/reference exists)")
    for pkg in ("toppra", "toppra.solverwrapper", "toppra.constraint"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # a package, with nothing to find in it
            sys.modules[pkg] = m
    sys.modules["toppra.constraint"].ConstraintType = _StandInConstraintType
    sys.modules["toppra"].constraint = sys.modules["toppra.constraint"]
    sys.modules["toppra"].solverwrapper = sys.modules["toppra.solverwrapper"]
    loader = importlib.machinery.ExtensionFileLoader(name, build_ref.so_path(name))
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    loader.exec_module(mod)
    _solver = mod
    return mod


def make_wrapper(constraints, path, gridpoints):
    """seidelWrapper as ReachabilityAlgorithm builds it (reachability_algorithm.py:121-125)."""
    return load().seidelWrapper(list(constraints), path, np.asarray(gridpoints, dtype=float), solve_lp1d=True)


def controllable_sets(w, sdmin, sdmax):
    """reachability_algorithm.py:166-238."""
    N = w.get_no_stages()
    K = np.zeros((N + 1, 2))
    K[N] = [sdmin ** 2, sdmax ** 2]
    g_upper = np.zeros(w.get_no_vars())
    g_upper[0], g_upper[1] = 1e-9, -1
    g_lower = -g_upper
    w.setup_solver()
    for i in range(N - 1, -1, -1):
        if np.isnan(K[i + 1]).any():
            K[i] = np.nan
        else:
            x_upper = w.solve_stagewise_optim(i, None, g_upper, np.nan, np.nan, K[i + 1, 0], K[i + 1, 1])[1]
            x_lower = w.solve_stagewise_optim(i, None, g_lower, np.nan, np.nan, K[i + 1, 0], K[i + 1, 1])[1]
            K[i] = [x_lower, x_upper]
        if K[i, 0] < 0:
            K[i, 0] = 0
        if np.isnan(K[i]).any():
            return K
    w.close_solver()
    return K


def parameterization(w, sd_start, sd_end):
    """reachability_algorithm.py:240-376 with TOPPRA._forward_step (time_optimal_algorithm.py:55-92).
    Returns (sdd, sd, K) with None, None on FailUncontrollable."""
    K = controllable_sets(w, sd_end, sd_end)
    if np.isnan(K).any():
        return None, None, K
    x_start = sd_start ** 2
    if x_start + SMALL < K[0, 0] or K[0, 1] + SMALL < x_start:
        return None, None, K
    N = w.get_no_stages()
    deltas = np.asarray(w.get_deltas())
    xs = np.zeros(N + 1)
    xs[0] = x_start
    us = np.zeros(N)
    g = np.zeros(w.get_no_vars())
    tries = 0
    w.setup_solver()
    i = 0
    while i < N:
        g[0], g[1] = -2 * deltas[i], -1
        res = w.solve_stagewise_optim(i, None, g, xs[i], xs[i], K[i + 1, 0], K[i + 1, 1])
        if np.isnan(res[0]):
            if tries < MAX_TRIES:
                xs[i] = max(xs[i] - TINY, 0.999 * xs[i])
                tries += 1
            else:
                xs[i + 1:] = np.nan
                break
        else:
            tries = 0
            us[i] = res[0]
            x_next = xs[i] + 2 * deltas[i] * us[i]
            x_next = max(x_next - TINY, 0.9999 * x_next)
            xs[i + 1] = min(K[i + 1, 1], max(K[i + 1, 0], x_next))
            i += 1
    w.close_solver()
    return us, np.sqrt(xs), K


def feasible_sets(w):
    """reachability_algorithm.py:131-164 (CVXPY_MAXX = 10000: constants.py)."""
    maxx = 10000.0
    N = w.get_no_stages()
    nV = w.get_no_vars()
    g_lower = np.zeros(nV)
    g_lower[0], g_lower[1] = 1e-9, 1
    X = np.zeros((N + 1, 2))
    w.setup_solver()
    for i in range(N + 1):
        X[i, 0] = w.solve_stagewise_optim(i, None, g_lower, -maxx, maxx, -maxx, maxx)[1]
        X[i, 1] = w.solve_stagewise_optim(i, None, -g_lower, -maxx, maxx, -maxx, maxx)[1]
    w.close_solver()
    X[X[:, 0] < 0, 0] = 0
    return X


class PrecomputedConstraint:
    """What seidelWrapper.__init__ asks of a constraint (cy_seidel_solverwrapper.pyx:437-455), answered from a stored tuple
    (picklable: the worker processes rebuild their wrappers from these without touching a GPU)."""

    def __init__(self, params, identical):
        self.params, self.identical = params, identical

    def get_constraint_type(self):
        load()
        return sys.modules["toppra.constraint"].ConstraintType.CanonicalLinear  # the real enum or the stand-in: whichever the solver compares with

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        return self.params


def constraint_tuples(coef, breaks, grid, vlim, alim):
    """The (a, b, c, F, g, ubound, xbound) tuples of JointVelocityConstraint and JointAccelerationConstraint (Interpolation)
    for ONE trajectory, on the CPU through the C restatement's q', q'' and velocity bound (linear_joint_velocity.py:43-53,
    linear_joint_acceleration.py:90-108, linear_constraint.py:84-192 for identical F)."""
    from . import oracle as orc
    grid = np.asarray(grid, dtype=float)
    d = coef.shape[-1]
    rows = orc.Wrapper(coef, breaks, grid, vlim, alim)  # its rows 2 .. 2 + d are q'(s_i), q''(s_i) (one C call per trajectory)
    qs, qss = rows.a_arr[:, 2:2 + d].copy(), rows.b_arr[:, 2:2 + d].copy()
    xb = np.zeros((len(grid), 2))
    for i in range(len(grid)):
        xb[i] = orc.velocity_xbound(qs[i], vlim)
    two_delta = 2 * np.diff(grid).reshape(-1, 1)
    a_next = np.concatenate((qs[1:] + two_delta * qss[1:], qs[-1:]), axis=0)
    b_next = np.concatenate((qss[1:], qss[-1:]), axis=0)
    a, b = np.hstack((qs, a_next)), np.hstack((qss, b_next))
    eye = np.vstack([np.eye(d), -np.eye(d)])
    g1 = np.concatenate([alim[:, 1], -alim[:, 0]])
    F = np.zeros((4 * d, 2 * d)); F[:2 * d, :d] = eye; F[2 * d:, d:] = eye
    vel = (None, None, None, None, None, None, xb)
    acc = (a, b, np.zeros_like(a), F, np.concatenate([g1, g1]), None, None)
    return vel, acc


def _pool_worker(jobs):
    """jobs: [(vel, acc, grid)] -- wrappers are built first (untimed: the reference pays for compute_constraint_params and
    the wrapper's row build once per trajectory too, but that is not the solver), then the passes are timed."""
    import time
    ws = [make_wrapper([PrecomputedConstraint(vel, False), PrecomputedConstraint(acc, True)], None, grid) for vel, acc, grid in jobs]
    t0 = time.perf_counter()
    ok = 0
    for w in ws:
        _, sd, _ = parameterization(w, 0.0, 0.0)
        ok += int(sd is not None and not np.isnan(sd).any())
    return time.perf_counter() - t0, ok, len(ws)


def time_passes(data, n, processes):
    """The first n trajectories of `data` (make_synthetic_batch layout) through the reference's compiled solver on
    `processes` worker processes (plain subprocesses of this module: no fork of a process that holds a GPU context, no
    dependence on the caller's __main__).  Returns dict(trajectories_per_s, ok, seconds = the slowest worker's pass
    time: setup -- constraint parameters, wrapper construction -- is outside it)."""
    import json
    import pickle
    import subprocess
    import tempfile
    jobs = [constraint_tuples(data["coef"][k], data["breaks"], data["grid"], data["vlim"][k], data["alim"][k]) + (np.asarray(data["grid"]),) for k in range(n)]
    chunks = [jobs[p::processes] for p in range(processes)]
    if processes == 1:
        res = [_pool_worker(chunks[0])]
    else:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory(prefix="tpr_refbase_") as tmp:
            procs = []
            for p, chunk in enumerate(chunks):
                path = os.path.join(tmp, "job%d.pkl" % p)
                with open(path, "wb") as fh:
                    pickle.dump(chunk, fh)
                env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")  # one core per worker
                procs.append(subprocess.Popen([sys.executable, "-m", "oracle.ref_solver_baseline", path], cwd=root, env=env,
                                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
            res = []
            for pr in procs:
                out, err = pr.communicate()
                if pr.returncode != 0:
                    raise RuntimeError("reference-solver worker failed: " + err[-400:])
                res.append(tuple(json.loads(out.strip().splitlines()[-1])))
    wall = max(r[0] for r in res)
    return {"trajectories_per_s": n / wall, "ok": sum(r[1] for r in res), "seconds": wall, "trajectories": n, "processes": processes}


if __name__ == "__main__":  # a worker of time_passes: python -m oracle.ref_solver_baseline <pickled jobs>
    import json
    import pickle
    with open(sys.argv[1], "rb") as fh:
        print(json.dumps(_pool_worker(pickle.load(fh))))

"""ctypes front-end of the CPU oracle (``oracle/seidel_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Nothing under ``toppra_amd/`` may import this module; it is used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the checker.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

FLAG_VEL = 1
FLAG_ACC = 2
FLAG_INTERP = 4
DEFAULT_FLAGS = FLAG_VEL | FLAG_ACC | FLAG_INTERP

OK, FAIL_UNCONTROLLABLE, ERR_UNKNOWN = 0, 1, 2


class LpSol(C.Structure):
    _fields_ = [("result", C.c_int), ("optval", C.c_double), ("optvar", C.c_double * 2),
                ("active_c", C.c_int * 2)]


_lib = None


def build(force=False):
    src = os.path.join(HERE, "seidel_oracle.c")
    if force or not os.path.exists(LIB) or (
            os.path.exists(src) and os.path.getmtime(LIB) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s", "liboracle.so"])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        lp = C.POINTER(C.c_long)
        L.orc_lp1d.restype = LpSol
        L.orc_lp1d.argtypes = [dp, C.c_int, dp, dp, C.c_double, C.c_double]
        L.orc_lp2d.restype = LpSol
        L.orc_lp2d.argtypes = [dp, C.c_int, dp, dp, dp, dp, dp, lp, lp, dp, dp]
        L.orc_path_eval.restype = None
        L.orc_path_eval.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, dp, dp]
        L.orc_velocity_xbound.restype = None
        L.orc_velocity_xbound.argtypes = [C.c_int, dp, dp, dp]
        L.orc_wrapper_new.restype = C.c_void_p
        L.orc_wrapper_new.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int]
        L.orc_wrapper_free.restype = None
        L.orc_wrapper_free.argtypes = [C.c_void_p]
        L.orc_wrapper_new_dense.restype = C.c_void_p
        L.orc_wrapper_new_dense.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int]
        L.orc_solve_stagewise_optim.restype = None
        L.orc_solve_stagewise_optim.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_double,
                                                C.c_double, C.c_double, dp]
        L.orc_compute_controllable_sets.restype = C.c_int
        L.orc_compute_controllable_sets.argtypes = [C.c_void_p, C.c_double, C.c_double, dp]
        L.orc_compute_feasible_sets.restype = None
        L.orc_compute_feasible_sets.argtypes = [C.c_void_p, dp]
        L.orc_compute_reachable_sets.restype = None
        L.orc_compute_reachable_sets.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, dp]
        L.orc_compute_parameterization.restype = C.c_int
        L.orc_compute_parameterization.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, dp, dp, dp]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, C.c_int,
                                      dp, dp, dp, dp, C.c_int, dp, dp, dp, C.POINTER(C.c_int32), C.c_int]
        L.orc_compute_parameterization_sd.restype = C.c_int
        L.orc_compute_parameterization_sd.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                                      dp, dp, dp, dp, dp]
        L.orc_solve_batch_sd.restype = C.c_int
        L.orc_solve_batch_sd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp, dp,
                                         dp, dp, dp, C.c_double, C.c_int, dp, dp, dp, C.POINTER(C.c_int32), dp,
                                         C.c_int]
        L.orc_robust_solve_batch.restype = C.c_int
        L.orc_robust_solve_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp, dp,
                                             dp, dp, dp, C.c_int, dp, dp, dp, dp, C.POINTER(C.c_int32), C.c_int]
        for name in ("a", "b", "c", "low", "high"):
            f = getattr(L, "orc_wrapper_" + name)
            f.restype = dp
            f.argtypes = [C.c_void_p]
        L.orc_wrapper_nC.restype = C.c_int
        L.orc_wrapper_nC.argtypes = [C.c_void_p]
        L.orc_wrapper_active.restype = None
        L.orc_wrapper_active.argtypes = [C.c_void_p, lp]
        L.orc_wrapper_set_active.restype = None
        L.orc_wrapper_set_active.argtypes = [C.c_void_p, lp]
        _lib = L
    return _lib


def _dp(x):
    return x.ctypes.data_as(C.POINTER(C.c_double)) if x is not None else None


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def lp1d(v, a, b, low, high):
    """Mirror of ``solve_lp1d`` (cy_seidel_solverwrapper.pyx:42-63)."""
    v = _f64(v)
    a = _f64([] if a is None else a)
    b = _f64([] if b is None else b)
    s = lib().orc_lp1d(_dp(v), len(a), _dp(a), _dp(b), float(low), float(high))
    return s.result, s.optval, s.optvar[0], s.active_c[0]


def lp2d(v, a, b, c, low, high, active_c):
    """Mirror of ``solve_lp2d`` (cy_seidel_solverwrapper.pyx:65-87)."""
    v = _f64(v)
    a = _f64([] if a is None else a)
    b = _f64([] if b is None else b)
    c = _f64([] if c is None else c)
    low, high = _f64(low), _f64(high)
    n = len(a)
    ac = np.ascontiguousarray(active_c, dtype=np.int64)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    a1 = np.zeros(n + 4)
    b1 = np.zeros(n + 4)
    s = lib().orc_lp2d(_dp(v), n, _dp(a), _dp(b), _dp(c), _dp(low), _dp(high),
                       ac.ctypes.data_as(C.POINTER(C.c_long)),
                       idx.ctypes.data_as(C.POINTER(C.c_long)), _dp(a1), _dp(b1))
    return s.result, s.optval, [s.optvar[0], s.optvar[1]], [s.active_c[0], s.active_c[1]]


def path_eval(coef, breaks, s):
    """q'(s), q''(s) for one trajectory; coef [4][nseg][d] (scipy CubicSpline.c)."""
    coef = _f64(coef)
    breaks = _f64(breaks)
    _, nseg, d = coef.shape
    qs = np.zeros(d)
    qss = np.zeros(d)
    lib().orc_path_eval(d, nseg, _dp(coef), _dp(breaks), float(s), _dp(qs), _dp(qss))
    return qs, qss


def velocity_xbound(qs, vlim):
    qs = _f64(qs)
    vlim = _f64(vlim)
    out = np.zeros(2)
    lib().orc_velocity_xbound(len(qs), _dp(qs), _dp(vlim), _dp(out))
    return out


class Wrapper:
    """One (constraints, path, grid) instance: the oracle's ``seidelWrapper`` + scan methods."""

    def __init__(self, coef, breaks, grid, vlim, alim, flags=DEFAULT_FLAGS, solve_lp1d=1):
        self.coef = _f64(coef)
        self.breaks = _f64(breaks)
        self.grid = _f64(grid)
        self.vlim = _f64(vlim) if vlim is not None else None
        self.alim = _f64(alim) if alim is not None else None
        _, self.nseg, self.d = self.coef.shape
        self.N = len(self.grid) - 1
        self._h = lib().orc_wrapper_new(self.d, self.nseg, self.N, _dp(self.coef), _dp(self.breaks),
                                        _dp(self.grid), _dp(self.vlim), _dp(self.alim), flags,
                                        solve_lp1d)
        self.nC = lib().orc_wrapper_nC(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_wrapper_free(self._h)
            self._h = None

    def _arr(self, name, cols):
        p = getattr(lib(), "orc_wrapper_" + name)(self._h)
        return np.ctypeslib.as_array(p, shape=(self.N + 1, cols)).copy()

    @property
    def a_arr(self):
        return self._arr("a", self.nC)

    @property
    def b_arr(self):
        return self._arr("b", self.nC)

    @property
    def c_arr(self):
        return self._arr("c", self.nC)

    @property
    def low_arr(self):
        return self._arr("low", 2)

    @property
    def high_arr(self):
        return self._arr("high", 2)

    def active(self):
        out = np.zeros(4, dtype=np.int64)
        lib().orc_wrapper_active(self._h, out.ctypes.data_as(C.POINTER(C.c_long)))
        return out

    def set_active(self, state):
        """Seed the warm-start state (active_c_up[0..1], active_c_down[0..1]) as earlier passes on the instance would have."""
        st = np.ascontiguousarray(state, dtype=np.int64)
        lib().orc_wrapper_set_active(self._h, st.ctypes.data_as(C.POINTER(C.c_long)))

    def solve_stagewise_optim(self, i, H, g, x_min, x_max, x_next_min, x_next_max):
        g = _f64(g)
        out = np.zeros(2)
        lib().orc_solve_stagewise_optim(self._h, int(i), _dp(g), x_min, x_max, x_next_min,
                                        x_next_max, _dp(out))
        return out

    def compute_controllable_sets(self, sdmin, sdmax):
        K = np.zeros((self.N + 1, 2))
        lib().orc_compute_controllable_sets(self._h, sdmin, sdmax, _dp(K))
        return K

    def compute_feasible_sets(self):
        X = np.zeros((self.N + 1, 2))
        lib().orc_compute_feasible_sets(self._h, _dp(X))
        return X

    def compute_reachable_sets(self, sdmin, sdmax):
        """Returns (L[N+1,2], X[N+1,2])."""
        L = np.zeros((self.N + 1, 2))
        X = np.zeros((self.N + 1, 2))
        lib().orc_compute_reachable_sets(self._h, float(sdmin), float(sdmax), _dp(L), _dp(X))
        return L, X

    def compute_parameterization(self, sd_start, sd_end):
        """Returns (status, sdd[N], sd[N+1], xs[N+1], K[N+1,2])."""
        sdd = np.zeros(self.N)
        sd = np.zeros(self.N + 1)
        xs = np.zeros(self.N + 1)
        K = np.zeros((self.N + 1, 2))
        st = lib().orc_compute_parameterization(self._h, sd_start, sd_end, _dp(sdd), _dp(sd),
                                                _dp(xs), _dp(K))
        return st, sdd, sd, xs, K


class DenseWrapper(Wrapper):
    """The oracle's ``seidelWrapper`` for ANY canonical-linear constraint list, from the arrays the reference's
    ``__init__`` ends up with (cy_seidel_solverwrapper.pyx:474-520): a, b, c [N+1, nC] (rows 0, 1 reserved), low, high
    [N+1, 2], deltas [N].  Scans as :class:`Wrapper`."""

    def __init__(self, a, b, c, low, high, deltas, solve_lp1d=1):
        a, b, c, low, high, deltas = (_f64(x) for x in (a, b, c, low, high, deltas))
        self.N, self.nC = a.shape[0] - 1, a.shape[1]
        assert b.shape == a.shape and c.shape == a.shape and low.shape == (self.N + 1, 2) and high.shape == (self.N + 1, 2)
        assert deltas.shape == (self.N,)
        self._h = lib().orc_wrapper_new_dense(self.N, self.nC, _dp(a), _dp(b), _dp(c), _dp(low), _dp(high), _dp(deltas),
                                              solve_lp1d)


def solve_dense_batch(a, b, c, low, high, deltas, sd_start=None, sd_end=None, want_X=False):
    """compute_parameterization (+ compute_feasible_sets) on dense rows for B trajectories, one fresh wrapper each:
    dict(sd2, sd, u, K, status[, X]) -- the checker of tpr_*_dense_batch."""
    a, b, c, low, high = (_f64(x) for x in (a, b, c, low, high))
    B, N1, _ = a.shape
    N = N1 - 1
    deltas = np.broadcast_to(_f64(deltas), (B, N))
    sd0 = np.broadcast_to(np.zeros(1) if sd_start is None else _f64(sd_start), (B,))
    sd1 = np.broadcast_to(np.zeros(1) if sd_end is None else _f64(sd_end), (B,))
    out = {"sd2": np.full((B, N + 1), np.nan), "sd": np.full((B, N + 1), np.nan), "u": np.full((B, N), np.nan),
           "K": np.zeros((B, N + 1, 2)), "status": np.zeros(B, dtype=np.int32)}
    if want_X:
        out["X"] = np.zeros((B, N + 1, 2))
    for i in range(B):
        if want_X:  # a fresh object, as compute_feasible_sets on a new instance
            out["X"][i] = DenseWrapper(a[i], b[i], c[i], low[i], high[i], np.ascontiguousarray(deltas[i])).compute_feasible_sets()
        w = DenseWrapper(a[i], b[i], c[i], low[i], high[i], np.ascontiguousarray(deltas[i]))
        st, sdd, sd, xs, K = w.compute_parameterization(float(sd0[i]), float(sd1[i]))
        out["status"][i] = st
        out["K"][i] = K
        if st != 1:
            out["sd2"][i], out["sd"][i], out["u"][i] = xs, sd, sdd
    return out


def solve_dense_batch_sd(a, b, c, low, high, deltas, desired_duration, sd_start=None, sd_end=None, atol=1e-5):
    """TOPPRAsd on dense rows for B trajectories (the checker of tpr_solve_desired_duration_dense_batch):
    dict(sd2, sd, u, K, status, alpha)."""
    a, b, c, low, high = (_f64(x) for x in (a, b, c, low, high))
    B, N1, _ = a.shape
    N = N1 - 1
    deltas = np.broadcast_to(_f64(deltas), (B, N))
    sd0 = np.broadcast_to(np.zeros(1) if sd_start is None else _f64(sd_start), (B,))
    sd1 = np.broadcast_to(np.zeros(1) if sd_end is None else _f64(sd_end), (B,))
    desired = np.broadcast_to(_f64(desired_duration), (B,))
    out = {"sd2": np.zeros((B, N + 1)), "sd": np.zeros((B, N + 1)), "u": np.zeros((B, N)), "K": np.zeros((B, N + 1, 2)),
           "status": np.zeros(B, dtype=np.int32), "alpha": np.zeros(B)}
    for i in range(B):
        w = DenseWrapper(a[i], b[i], c[i], low[i], high[i], np.ascontiguousarray(deltas[i]))
        alpha = C.c_double(float("nan"))
        sdd, sd, xs, K = np.zeros(N), np.zeros(N + 1), np.zeros(N + 1), np.zeros((N + 1, 2))
        out["status"][i] = lib().orc_compute_parameterization_sd(w._h, float(sd0[i]), float(sd1[i]), float(desired[i]), float(atol),
                                                                 _dp(sdd), _dp(sd), _dp(xs), _dp(K), C.byref(alpha))
        out["u"][i], out["sd"][i], out["sd2"][i], out["K"][i], out["alpha"][i] = sdd, sd, xs, K, alpha.value
    return out


def robust_solve_batch(coef, breaks, grid, vlim, alim, ellipsoid, sd_start=None, sd_end=None,
                       flags=DEFAULT_FLAGS, want_X=True, nthreads=1):
    """Robust (conic) TOPP-RA, PARITY UNPINNED against ECOS, cross-checked at 1e-7 against robust_independent.py (see seidel_oracle.c): dict(sd2, u, K, X, status)."""
    coef, breaks, grid = _f64(coef), _f64(breaks), _f64(grid)
    B, _, nseg, d = coef.shape
    N = grid.shape[-1] - 1
    vlim = _f64(vlim) if vlim is not None else None
    alim = _f64(alim)
    ell = _f64(ellipsoid)
    sd_start = _f64(sd_start) if sd_start is not None else None
    sd_end = _f64(sd_end) if sd_end is not None else None
    sd2, u, K = np.zeros((B, N + 1)), np.zeros((B, N)), np.zeros((B, N + 1, 2))
    X = np.zeros((B, N + 1, 2)) if want_X else None
    status = np.zeros(B, dtype=np.int32)
    if vlim is None:
        flags &= ~FLAG_VEL
    lib().orc_robust_solve_batch(B, d, nseg, N, _dp(coef), _dp(breaks), int(breaks.ndim == 2), _dp(grid),
                                 int(grid.ndim == 2), _dp(vlim), _dp(alim), _dp(ell), _dp(sd_start), _dp(sd_end),
                                 flags, _dp(sd2), _dp(u), _dp(K), _dp(X), status.ctypes.data_as(C.POINTER(C.c_int32)),
                                 int(nthreads))
    return {"sd2": sd2, "u": u, "K": K, "X": X, "status": status}


def solve_batch_sd(coef, breaks, grid, vlim, alim, desired_duration, sd_start=None, sd_end=None, atol=1e-5,
                   flags=DEFAULT_FLAGS, nthreads=1):
    """TOPPRAsd batch driver: dict(sd2, u, K, status, alpha)."""
    coef, breaks, grid = _f64(coef), _f64(breaks), _f64(grid)
    B, _, nseg, d = coef.shape
    N = grid.shape[-1] - 1
    vlim = _f64(vlim) if vlim is not None else None
    alim = _f64(alim) if alim is not None else None
    sd_start = _f64(sd_start) if sd_start is not None else None
    sd_end = _f64(sd_end) if sd_end is not None else None
    desired = _f64(np.broadcast_to(desired_duration, (B,)))
    sd2, u, K = np.zeros((B, N + 1)), np.zeros((B, N)), np.zeros((B, N + 1, 2))
    status, alpha = np.zeros(B, dtype=np.int32), np.zeros(B)
    lib().orc_solve_batch_sd(B, d, nseg, N, _dp(coef), _dp(breaks), int(breaks.ndim == 2), _dp(grid),
                             int(grid.ndim == 2), _dp(vlim), _dp(alim), _dp(sd_start), _dp(sd_end), _dp(desired),
                             float(atol), flags, _dp(sd2), _dp(u), _dp(K), status.ctypes.data_as(C.POINTER(C.c_int32)),
                             _dp(alpha), int(nthreads))
    return {"sd2": sd2, "u": u, "K": K, "status": status, "alpha": alpha}


def solve_batch(coef, breaks, grid, vlim, alim, sd_start=None, sd_end=None, flags=DEFAULT_FLAGS,
                nthreads=1):
    """Batch driver: coef [B][4][nseg][d]; breaks [nseg+1] or [B][nseg+1]; grid [N+1] or [B][N+1].

    Returns dict(sd2[B][N+1], u[B][N], K[B][N+1][2], status[B])."""
    coef = _f64(coef)
    breaks = _f64(breaks)
    grid = _f64(grid)
    B, _, nseg, d = coef.shape
    N = grid.shape[-1] - 1
    vlim = _f64(vlim) if vlim is not None else None
    alim = _f64(alim) if alim is not None else None
    sd_start = _f64(sd_start) if sd_start is not None else None
    sd_end = _f64(sd_end) if sd_end is not None else None
    sd2 = np.zeros((B, N + 1))
    u = np.zeros((B, N))
    K = np.zeros((B, N + 1, 2))
    status = np.zeros(B, dtype=np.int32)
    lib().orc_solve_batch(B, d, nseg, N, _dp(coef), _dp(breaks), int(breaks.ndim == 2), _dp(grid),
                          int(grid.ndim == 2), _dp(vlim), _dp(alim), _dp(sd_start), _dp(sd_end), flags,
                          _dp(sd2), _dp(u), _dp(K), status.ctypes.data_as(C.POINTER(C.c_int32)),
                          int(nthreads))
    return {"sd2": sd2, "u": u, "K": K, "status": status}

"""Solver wrapper of the HIP path: the reference's ``SolverWrapper`` contract
(solverwrapper/solverwrapper.py:49-166) with ``seidelWrapper``'s concrete signatures
(cy_seidel_solverwrapper.pyx:425-544, :549-703).

``hipSeidelWrapper`` owns one (constraints, path, grid) problem.  The pass-level methods
(``controllable_sets`` / ``parameterization`` / ``feasible_sets``) are what the algorithm layer
calls -- one kernel launch per pass instead of 3N Python->solver round trips.
``solve_stagewise_optim`` is kept as the single-LP compatibility entry with the reference's
stateful warm start (``active_c_up`` / ``active_c_down`` persist between calls).
"""
import ctypes as C

import numpy as np

from . import _capi, batch
from .interpolator import spline_tables


def available_solvers(output_msg=True):
    """Mirror of solverwrapper.available_solvers(): name/availability pairs, best first."""
    try:
        ok = _capi.device_count() > 0
    except Exception:
        ok = False
    avail = (("hip", ok),)
    if output_msg:
        print(avail)
    return avail


def extract_limits(constraint_list, dof):
    """(vlim, alim, interpolation) from a constraint list -- ours or the reference's classes
    (duck-typed on .vlim / .alim / .discretization_type).  Anything else is refused the way
    seidelWrapper refuses non-canonical-linear constraints (cy_seidel_solverwrapper.pyx:459-460)."""
    vlim = alim = None
    interpolation = True
    for c in constraint_list:
        ctype = getattr(c.get_constraint_type(), "value", None)
        if ctype != 0:
            raise NotImplementedError("the seidel path handles CanonicalLinear constraints only")
        if hasattr(c, "vlim") and not hasattr(c, "vlim_func"):
            if vlim is not None:
                raise NotImplementedError("more than one JointVelocityConstraint")
            vlim = np.ascontiguousarray(c.vlim, dtype=np.float64)
        elif hasattr(c, "alim"):
            if alim is not None:
                raise NotImplementedError("more than one JointAccelerationConstraint")
            alim = np.ascontiguousarray(c.alim, dtype=np.float64)
            interpolation = getattr(c.get_discretization_type(), "value", 1) == 1
        else:
            raise NotImplementedError(
                "%s is outside the HIP path (JointVelocityConstraint and JointAccelerationConstraint "
                "are supported)" % type(c).__name__)
        if c.get_dof() != dof:
            raise ValueError("Wrong dimension: constraint dof ({:d}) not equal to path dof ({:d})".format(
                c.get_dof(), dof))
    return vlim, alim, interpolation


class SolverWrapper(object):
    """Interface (solverwrapper.py:49-166)."""

    def get_no_stages(self):
        return self.N

    def get_no_vars(self):
        return self.nV

    def get_deltas(self):
        return self.deltas

    def solve_stagewise_optim(self, i, H, g, x_min, x_max, x_next_min, x_next_max):
        raise NotImplementedError

    def setup_solver(self):
        pass

    def close_solver(self):
        pass


class hipSeidelWrapper(SolverWrapper):
    """Seidel LP wrapper running on the MI355X.

    Parameters as seidelWrapper: ``constraint_list``, ``path``, ``path_discretization``,
    ``solve_lp1d`` (solve the 1-variable LP when x_min == x_max)."""

    def __init__(self, constraint_list, path, path_discretization, solve_lp1d=0):
        self.constraints = constraint_list
        self.path = path
        self.path_discretization = np.array(path_discretization, dtype=np.float64)
        self.N = len(self.path_discretization) - 1
        self.deltas = self.path_discretization[1:] - self.path_discretization[:-1]
        self.nV = 2
        self._solve_lp1d = int(solve_lp1d)
        coef, breaks = spline_tables(path)
        self._coef, self._breaks = coef[None], breaks
        self.dof = coef.shape[2]
        vlim, alim, self._interp = extract_limits(constraint_list, self.dof)
        self._vlim = None if vlim is None else vlim[None]
        self._alim = None if alim is None else alim[None]
        self.nC = 2 + (0 if alim is None else (4 if self._interp else 2) * self.dof)
        # warm-start state of the two LP "solvers", as in the reference object
        self._active = np.zeros((1, 4), dtype=np.int32)
        self._params = None  # device init is lazy: every compute entry calls _capi.init()
        self._call_state = None  # persistent problem / result structures of `parameterization`

    @property
    def params(self):
        if self._params is None:
            self._params = [c.compute_constraint_params(self.path, self.path_discretization)
                            for c in self.constraints]
        return self._params

    # -- pass-level entries ------------------------------------------------------------------
    def _args(self):
        return self._coef, self._breaks, self.path_discretization, self._vlim, self._alim

    # (every pass reads and updates self._active, the reference object's warm-start state: a sequence of passes on
    # one instance -- examples/plot_kinematics.py:48,72 -- pivots in the reference's order and returns its bits)
    # Boundary velocities are squared HERE, with the reference's own expression on the caller's own objects
    # (`sd ** 2`: libm pow for Python floats, which is one ulp off sd * sd now and then; numpy's square for numpy
    # scalars), and handed down as x = sd^2 (TPR_BOUNDARY_SQUARED).
    def controllable_sets(self, sdmin, sdmax):
        return batch.controllable_sets_batch(*self._args(), np.array([sdmin ** 2], dtype=np.float64),
                                             np.array([sdmax ** 2], dtype=np.float64), self._interp, active=self._active,
                                             squared=True)[0]

    def feasible_sets(self):
        return batch.feasible_sets_batch(*self._args(), self._interp, active=self._active)[0]

    def reachable_sets(self, sdmin, sdmax):
        L, X = batch.reachable_sets_batch(*self._args(), np.array([sdmin ** 2], dtype=np.float64),
                                          np.array([sdmax ** 2], dtype=np.float64), self._interp, want_X=True, squared=True)
        return L[0], X[0]

    def parameterization(self, sd_start, sd_end):
        """One compute_parameterization: sd2, sd, u, K [copies of the per-instance result buffers], status.  A single
        trajectory is pure call latency, so the problem description, the result buffers and the ctypes
        arguments are built once per instance; a call writes the two boundary velocities and makes one
        library call (which takes its small-host-call path: csrc/tpr_kernels.hip)."""
        st = self._call_state
        if st is None:
            sd0, sd1 = np.zeros(1), np.zeros(1)
            p, keep = _capi.make_problem(*self._args(), sd0, sd1, self._interp, active=self._active, squared=True)
            N = self.N
            out = {"sd2": np.empty((1, N + 1)), "sd": np.empty((1, N + 1)), "u": np.empty((1, N)),
                   "K": np.empty((1, N + 1, 2)), "status": np.empty((1,), dtype=np.int32)}
            r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out["sd"]), u=_capi.ptr(out["u"]),
                                 K=_capi.ptr(out["K"]), status=_capi.ptr(out["status"]))
            views = {k: v[0] for k, v in out.items() if k != "status"}
            st = self._call_state = (sd0, sd1, C.byref(p), C.byref(r), _capi.load().tpr_solve_batch, views,
                                     out["status"], (p, r, keep, out))
        sd0, sd1, pref, rref, fn, views, status = st[:7]
        sd0[0] = sd_start ** 2
        sd1[0] = sd_end ** 2
        _capi.init()
        rc = fn(pref, rref, None)
        if rc != 0:
            _capi.check(rc)
        res = {k: v.copy() for k, v in views.items()}  # (the buffers are reused by the next call: never hand out views)
        res["status"] = int(status[0])
        return res

    def parameterization_sd(self, sd_start, sd_end, desired_duration, atol=1e-5):
        out = batch.solve_desired_duration_batch(*self._args(), desired_duration,
                                                 np.array([sd_start ** 2], dtype=np.float64),
                                                 np.array([sd_end ** 2], dtype=np.float64), atol, squared=True)
        return {k: v[0] for k, v in out.items()}

    # -- single-LP compatibility entry ---------------------------------------------------------
    def solve_stagewise_optim(self, i, H, g, x_min, x_max, x_next_min, x_next_max):
        assert 0 <= i <= self.N
        _capi.init()
        p, keep = _capi.make_problem(*self._args(), None, None, self._interp)
        stage = np.array([i], dtype=np.int32)
        g = np.ascontiguousarray(np.asarray(g, dtype=np.float64)[:2].reshape(1, 2))
        xb = np.array([[x_min, x_max, x_next_min, x_next_max]], dtype=np.float64)
        out = np.empty((1, 2))
        _capi.check(_capi.load().tpr_solve_stagewise_batch(
            C.byref(p), _capi.ptr(stage), _capi.ptr(g), _capi.ptr(xb), _capi.ptr(self._active),
            self._solve_lp1d, _capi.ptr(out), None))
        return out[0]


def dense_rows(constraint_list, path, path_discretization):
    """The arrays seidelWrapper.__init__ builds from ANY list of canonical-linear constraints
    (cy_seidel_solverwrapper.pyx:455-520): dict(a, b, c [N+1, nC] -- rows 0, 1 reserved for the x_next pair, then one
    block of rows F a, F b, F c - g per constraint --, low, high [N+1, 2] -- the +-1e8 variable box tightened by the
    constraints' ubound / xbound --, deltas [N], params).  Host numpy with the reference's own calls (``a.dot(F.T)``
    for identical constraints, one ``F_i.dot(a_i)`` per gridpoint otherwise), so the rows are the reference's bits for
    the same numpy; the parameters themselves come from the constraints' (user) callbacks."""
    grid = np.array(path_discretization, dtype=np.float64)
    n1 = len(grid)
    params, blocks = [], []
    low, high = np.full((n1, 2), -1e8), np.full((n1, 2), 1e8)  # VAR_MIN / VAR_MAX (:22-23)
    for con in constraint_list:
        if getattr(con.get_constraint_type(), "value", None) != 0:
            raise NotImplementedError("the seidel path handles CanonicalLinear constraints only")
        a, b, c, F, g, ubound, xbound = con.compute_constraint_params(path, grid)
        params.append((a, b, c, F, g, ubound, xbound))
        if a is not None:
            if getattr(con, "identical", False):
                blocks.append((a.dot(F.T), b.dot(F.T), c.dot(F.T) - g))
            else:
                rows = [(np.dot(F[i], a[i]), np.dot(F[i], b[i]), np.dot(F[i], c[i]) - g[i]) for i in range(n1)]
                blocks.append(tuple(np.array([r[k] for r in rows]) for k in range(3)))
        for col, bound in ((0, ubound), (1, xbound)):
            if bound is not None:  # dbl_max(a, b) = a if a > b else b (:43-50)
                low[:, col] = np.where(low[:, col] > bound[:, 0], low[:, col], bound[:, 0])
                high[:, col] = np.where(high[:, col] < bound[:, 1], high[:, col], bound[:, 1])
    nC = 2 + sum(blk[0].shape[1] for blk in blocks)
    out = {k: np.zeros((n1, nC)) for k in ("a", "b", "c")}
    col = 2
    for blk in blocks:
        m = blk[0].shape[1]
        for k, arr in zip(("a", "b", "c"), blk):
            out[k][:, col:col + m] = arr
        col += m
    out.update(low=low, high=high, deltas=grid[1:] - grid[:-1], nC=nC, params=params)
    return out


class hipDenseSeidelWrapper(SolverWrapper):
    """seidelWrapper for ANY list of canonical-linear constraints (SecondOrderConstraint, JointTorqueConstraint, the
    reference's own constraint objects, hand-written ones): the constraints' parameters are evaluated on the host, as in
    the reference, flattened by :func:`dense_rows`, and the passes run on the dense-row entries of the library
    (tpr_*_dense_batch: rows across lanes, the reference's full Seidel iteration).  The object's warm-start state is carried
    from pass to pass as the reference object carries it.
    TOPPRAsd and reachable sets run here too."""

    def __init__(self, constraint_list, path, path_discretization, solve_lp1d=1):
        self.constraints = constraint_list
        self.path = path
        self.path_discretization = np.array(path_discretization, dtype=np.float64)
        self.N = len(self.path_discretization) - 1
        self.nV = 2
        for c in constraint_list:
            if c.get_dof() != path.dof:
                raise ValueError("Wrong dimension: constraint dof ({:d}) not equal to path dof ({:d})".format(
                    c.get_dof(), path.dof))
        rows = dense_rows(constraint_list, path, self.path_discretization)
        self.deltas = rows["deltas"]
        self.nC = rows["nC"]
        self.params = rows["params"]
        if self.nC > 122:
            raise NotImplementedError("%d constraint rows per stage: the dense-row kernels hold 122" % self.nC)
        self._rows = tuple(np.ascontiguousarray(rows[k][None]) for k in ("a", "b", "c", "low", "high")) + (self.deltas,)
        self._solve_lp1d = int(solve_lp1d)
        # warm-start state of the two LP "solvers", as in the reference object: every pass and every per-stage call reads
        # and updates it (tpr_dense_problem.active), so a sequence of calls on ONE instance pivots in the reference's order
        self._active = np.zeros((1, 4), dtype=np.int32)
        self._active_up, self._active_down = self._active[0, 0:2], self._active[0, 2:4]

    def controllable_sets(self, sdmin, sdmax):
        return batch.controllable_sets_dense_batch(*self._rows, np.array([sdmin ** 2], dtype=np.float64),
                                                   np.array([sdmax ** 2], dtype=np.float64), squared=True, active=self._active)[0]

    def feasible_sets(self):
        return batch.feasible_sets_dense_batch(*self._rows, active=self._active)[0]

    def parameterization(self, sd_start, sd_end):
        out = batch.solve_dense_batch(*self._rows, np.array([sd_start ** 2], dtype=np.float64),
                                      np.array([sd_end ** 2], dtype=np.float64), want_sd=True, squared=True, active=self._active)
        res = {k: v[0] for k, v in out.items() if k != "status"}
        res["status"] = int(out["status"][0])
        return res

    # -- single-LP compatibility entry (cy_seidel_solverwrapper.pyx:549-697) on the stage's dense rows: the bounds and the
    # x_next pair are set up as the reference sets them up, the LP itself runs on the library's LP entries
    # (tpr_lp1d_batch / tpr_lp2d_batch = cy_solve_lp1d / cy_solve_lp2d), the two warm-start sets persist between calls
    def solve_stagewise_optim(self, i, H, g, x_min, x_max, x_next_min, x_next_max):
        assert 0 <= i <= self.N
        _capi.init()
        lib = _capi.load()
        a, b, c = (np.array(r[0, i]) for r in self._rows[:3])
        low, high = np.array(self._rows[3][0, i]), np.array(self._rows[4][0, i])
        if not np.isnan(x_min):
            low[1] = low[1] if low[1] > x_min else x_min
        if not np.isnan(x_max):
            high[1] = high[1] if high[1] < x_max else x_max
        a[:2], b[:2], c[:2] = 0.0, 0.0, -1.0  # absent bounds / the last stage: the disabled row
        if i < self.N:
            if not np.isnan(x_next_min):
                a[0], b[0], c[0] = -2 * self.deltas[i], -1.0, x_next_min
            if not np.isnan(x_next_max):
                a[1], b[1], c[1] = 2 * self.deltas[i], 1.0, -x_next_max
        upper = g[1] > 0
        res, val, active = np.zeros(1, np.int32), np.zeros(1), np.zeros(2, np.int32)
        if x_min == x_max and self._solve_lp1d > 0:
            bx_c = b * x_min + c
            v = np.array([-g[0], -g[1] * x_min], dtype=np.float64)
            var, u_low, u_high = np.zeros(1), np.array([low[0]]), np.array([high[0]])  # (named: the call takes raw addresses)
            _capi.check(lib.tpr_lp1d_batch(1, self.nC, _capi.ptr(v), _capi.ptr(a), _capi.ptr(bx_c), _capi.ptr(u_low),
                                           _capi.ptr(u_high), _capi.ptr(res), _capi.ptr(val), _capi.ptr(var),
                                           _capi.ptr(active), None))
            if res[0] == 0:
                return np.array([np.nan, np.nan])
            (self._active_up if upper else self._active_down)[0] = active[0]
            return np.array([var[0], x_min])
        warm = self._active_up if upper else self._active_down
        v = np.array([-g[0], -g[1], 0.0], dtype=np.float64)
        var = np.zeros(2)
        _capi.check(lib.tpr_lp2d_batch(1, self.nC, _capi.ptr(v), _capi.ptr(a), _capi.ptr(b), _capi.ptr(c), _capi.ptr(low),
                                       _capi.ptr(high), _capi.ptr(warm), _capi.ptr(res), _capi.ptr(val), _capi.ptr(var),
                                       _capi.ptr(active), None))
        if res[0] == 0:
            return np.array([np.nan, np.nan])
        warm[:] = active
        return var

    def reachable_sets(self, sdmin, sdmax):
        L, X = batch.reachable_sets_dense_batch(*self._rows, np.array([sdmin ** 2], dtype=np.float64),
                                                np.array([sdmax ** 2], dtype=np.float64), want_X=True, active=self._active,
                                                squared=True)
        return L[0], X[0]

    def parameterization_sd(self, sd_start, sd_end, desired_duration, atol=1e-5):
        out = batch.solve_desired_duration_dense_batch(*self._rows, desired_duration, np.array([sd_start ** 2], dtype=np.float64),
                                                       np.array([sd_end ** 2], dtype=np.float64), atol, active=self._active,
                                                       squared=True)
        return {k: v[0] for k, v in out.items()}


class hipRobustWrapper(SolverWrapper):
    """Wrapper for [JointVelocityConstraint (optional), RobustLinearConstraint(JointAccelerationConstraint)]
    problems -- the role ``ecosWrapper`` plays in the reference (ecos_solverwrapper.py:14-207).

    PARITY UNPINNED against ECOS (absent here; the reference holds no golden vectors for it), cross-checked at 1e-7
    against an independent exact solver (tests/test_gpu_robust.py): the stage problems are the ones the reference builds for ECOS, solved exactly
    on the GPU (csrc/tpr_robust.hip.inc) instead of by ECOS's interior-point iteration."""

    def __init__(self, constraint_list, path, path_discretization):
        self.constraints = constraint_list
        self.path = path
        self.path_discretization = np.array(path_discretization, dtype=np.float64)
        self.N = len(self.path_discretization) - 1
        self.deltas = self.path_discretization[1:] - self.path_discretization[:-1]
        self.nV = 2
        coef, breaks = spline_tables(path)
        self._coef, self._breaks = coef[None], breaks
        self.dof = coef.shape[2]
        self._vlim = self._alim = self._ell = None
        self._interp = True
        for c in constraint_list:
            ctype = getattr(c.get_constraint_type(), "value", None)
            if ctype == 1 and hasattr(c, "base_constraint") and hasattr(c.base_constraint, "alim"):
                if self._alim is not None:
                    raise NotImplementedError("more than one robust constraint")
                self._alim = np.ascontiguousarray(c.base_constraint.alim, dtype=np.float64)[None]
                self._ell = np.asarray(c.ellipsoid_axes_lengths, dtype=np.float64).reshape(3)
                self._interp = getattr(c.get_discretization_type(), "value", 0) == 1
            elif ctype == 0 and hasattr(c, "vlim") and not hasattr(c, "vlim_func"):
                self._vlim = np.ascontiguousarray(c.vlim, dtype=np.float64)[None]
            else:
                raise NotImplementedError("%s is outside the robust HIP path" % type(c).__name__)
            if c.get_dof() != self.dof:
                raise ValueError("Wrong dimension: constraint dof ({:d}) not equal to path dof ({:d})".format(
                    c.get_dof(), self.dof))
        if self._alim is None:
            raise NotImplementedError("the robust path needs a RobustLinearConstraint over JointAccelerationConstraint")

    def _solve(self, sd_start, sd_end, want_X=False):
        out = batch.robust_solve_batch(self._coef, self._breaks, self.path_discretization, self._vlim, self._alim,
                                       self._ell, np.array([sd_start], dtype=np.float64),
                                       np.array([sd_end], dtype=np.float64), self._interp, want_X=want_X)
        return {k: v[0] for k, v in out.items()}

    def parameterization(self, sd_start, sd_end):
        return self._solve(sd_start, sd_end)

    def controllable_sets(self, sdmin, sdmax):
        if sdmin != sdmax:
            raise NotImplementedError("robust controllable sets are implemented for sdmin == sdmax")
        return self._solve(sdmin, sdmax)["K"]

    def feasible_sets(self):
        return self._solve(0.0, 0.0, want_X=True)["X"]

    def solve_stagewise_optim(self, i, H, g, x_min, x_max, x_next_min, x_next_max):
        raise NotImplementedError("the robust wrapper works at the pass level only")

"""Independent checker for the robust (conic) TOPP-RA stage problems -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under
oracle/; the product (toppra_amd/) never does.

The reference hands every stage of a ``RobustLinearConstraint`` problem to ECOS
(toppra/solverwrapper/ecos_solverwrapper.py:90-207), which is not installed here.  The product
(csrc/tpr_robust.hip.inc) and its CPU restatement (seidel_oracle.c, ``orc_robust_*``) both solve those
stage problems by one method: a closed-form u-interval per row at fixed x, then bisection on x.  This
module solves the *same problems by a different method*, so that agreement is evidence and not a
tautology:

  * the problem is rebuilt from the reference's definition, from scipy/numpy only: rows
    ``a u + b x + c + ||(ru u, rx x, rc)||_2 <= 0`` (conic_constraint.py:19-26, :95-124) over the rows of
    JointAccelerationConstraint (linear_joint_acceleration.py:63-104, Interpolation transform
    linear_constraint.py:164-190); the velocity box (fp32 arithmetic of _CythonUtils.pyx:16-59) capped at
    ECOS_MAXX (ecos_solverwrapper.py:172); absent bounds replaced by +-ECOS_INFTY (:112-135); the
    ``x + 2 delta u`` bounds (:124-135);
  * it is solved by Kelley's cutting-plane method: the cone rows are outer-linearised by tangent planes
    at the current optimum, and every linear programme in (u, x) is solved by exhaustive enumeration of
    the pairwise intersections of its rows.  No closed-form root of a cone row, no bisection.

Accuracy: the iteration stops when every cone row holds to 3e-12 (relative to the row's size) at the
LP optimum (the LP vertices themselves are feasible to 1e-13), which pins the optimal x to ~1e-10; the tests compare at 1e-7 (K, X) as stated in
DESIGN.md section 7.
"""
import numpy as np
from scipy.interpolate import PPoly

ECOS_INFTY = 1000.0   # toppra/constants.py:47
ECOS_MAXX = 10000.0   # toppra/constants.py:46
FEAS_MAXX = 10000.0   # toppra/constants.py:42 (compute_feasible_sets bounds, reachability_algorithm.py:151-156)
U_BOX = 1.0e7         # u is unbounded in the reference; the rows keep |u| far below this


def _lp2(A, h, cost):
    """max cost.z  s.t.  A z <= h  (z in R^2) by enumerating all pairwise row intersections.
    Returns (z, value) or (None, None) when no vertex is feasible."""
    m = A.shape[0]
    i, j = np.triu_indices(m, 1)
    a1, b1, a2, b2 = A[i, 0], A[i, 1], A[j, 0], A[j, 1]
    det = a1 * b2 - a2 * b1
    scale = np.hypot(a1, b1) * np.hypot(a2, b2)
    ok = np.abs(det) > 1e-13 * scale
    i, j, det = i[ok], j[ok], det[ok]
    zu = (h[i] * A[j, 1] - h[j] * A[i, 1]) / det
    zx = (A[i, 0] * h[j] - A[j, 0] * h[i]) / det
    Z = np.stack([zu, zx], axis=1)
    res = Z @ A.T - h[None, :]
    tol = 1e-13 * (1.0 + np.abs(h))[None, :] + 1e-13 * (np.abs(Z) @ np.abs(A.T))
    feas = np.all(res <= tol, axis=1)
    if not feas.any():
        return None, None
    Z = Z[feas]
    val = Z @ cost
    k = int(np.argmax(val))
    return Z[k], float(val[k])


def _cone_value_grad(a, b, c, ell, z):
    u, x = z
    nrm = np.sqrt((ell[0] * u) ** 2 + (ell[1] * x) ** 2 + ell[2] ** 2)
    f = a * u + b * x + c + nrm
    if nrm > 0:
        gu = a + ell[0] ** 2 * u / nrm
        gx = b + ell[1] ** 2 * x / nrm
    else:  # ||.|| is not differentiable at the origin when rc = 0: any subgradient will do
        gu, gx = a, b
    return f, gu, gx


def solve_stage(a, b, c, ell, lin_A, lin_h, cost, max_iter=200):
    """max cost.(u, x) over { cone rows (a, b, c, ell) } and { lin_A z <= lin_h }.
    Returns (u, x) or None when infeasible."""
    a, b, c = (np.asarray(v, dtype=np.float64) for v in (a, b, c))
    A = [np.asarray(lin_A, dtype=np.float64), np.array([[1.0, 0.0], [-1.0, 0.0]])]
    h = [np.asarray(lin_h, dtype=np.float64), np.array([U_BOX, U_BOX])]
    # initial outer approximation: ||.|| >= |rc| makes  a u + b x + c + |rc| <= 0  a valid cut for every row
    A.append(np.stack([a, b], axis=1))
    h.append(-(c + abs(ell[2])))
    A, h = np.concatenate(A), np.concatenate(h)
    cost = np.asarray(cost, dtype=np.float64)
    for _ in range(max_iter):
        z, _ = _lp2(A, h, cost)
        if z is None:
            return None
        f, gu, gx = _cone_value_grad(a, b, c, ell, z)
        size = np.abs(a * z[0]) + np.abs(b * z[1]) + np.abs(c) + 1.0
        bad = f > 3e-12 * size
        if not bad.any():
            return z
        # tangent cuts of the violated rows at z:  f(z) + g.(w - z) <= 0
        cutA = np.stack([gu[bad], gx[bad]], axis=1)
        cuth = cutA @ z - f[bad]
        A, h = np.concatenate([A, cutA]), np.concatenate([h, cuth])
    raise RuntimeError("cutting planes did not converge")


# ---------------------------------------------------------------------------------------------------
# the stage problems of one trajectory, rebuilt from the reference's definition

class RobustTrajectory:
    """One path + limits + ellipsoid; builds every stage problem the reference's ecosWrapper would."""

    def __init__(self, coef, breaks, grid, vlim, alim, ell, interpolation=True):
        coef = np.asarray(coef, dtype=np.float64)  # [4][nseg][d]
        self.grid = np.asarray(grid, dtype=np.float64)
        self.N = len(self.grid) - 1
        pp = PPoly(coef, np.asarray(breaks, dtype=np.float64))
        self.qs = pp.derivative()(self.grid)       # [N+1][d], as interpolator.py:423-430
        self.qss = pp.derivative(2)(self.grid)
        self.vlim = None if vlim is None else np.asarray(vlim, dtype=np.float64)
        self.alim = np.asarray(alim, dtype=np.float64)
        self.ell = np.asarray(ell, dtype=np.float64)
        self.interp = bool(interpolation)

    def xbound(self, i):
        """JointVelocityConstraint's bound on x (fp32, _CythonUtils.pyx:16-59)."""
        if self.vlim is None:
            return None
        sdmin, sdmax = np.float32(-1e8), np.float32(1e8)
        for k, q in enumerate(self.qs[i]):
            if q > 0:
                sdmax = np.float32(min(self.vlim[k, 1] / q, float(sdmax)))
                sdmin = np.float32(max(self.vlim[k, 0] / q, float(sdmin)))
            elif q < 0:
                sdmax = np.float32(min(self.vlim[k, 0] / q, float(sdmax)))
                sdmin = np.float32(max(self.vlim[k, 1] / q, float(sdmin)))
        lo = max(float(sdmin), 0.0)
        return lo * lo, float(np.float32(sdmax) * np.float32(sdmax))

    def rows(self, i):
        """(a, b, c) of the conic rows at stage i: F a, F b, F c - g with F = [I; -I] per block."""
        amax, amin = self.alim[:, 1], self.alim[:, 0]
        q1, q2 = self.qs[i], self.qss[i]
        a = [q1, -q1]
        b = [q2, -q2]
        c = [-amax, amin]
        if self.interp:
            if i < self.N:
                delta = self.grid[i + 1] - self.grid[i]
                an = self.qs[i + 1] + 2 * delta * self.qss[i + 1]
                bn = self.qss[i + 1]
            else:  # the last gridpoint repeats its own rows (linear_constraint.py:172,177)
                an, bn = q1, q2
            a += [an, -an]
            b += [bn, -bn]
            c += [-amax, amin]
        return np.concatenate(a), np.concatenate(b), np.concatenate(c)

    def linear(self, i, x_min, x_max, x_next_min, x_next_max):
        """The linear part of ecosWrapper.solve_stagewise_optim's G, h (ecos_solverwrapper.py:108-175)."""
        nan = np.isnan
        A = [[0.0, -1.0], [0.0, 1.0]]
        h = [ECOS_INFTY if nan(x_min) else -x_min, ECOS_INFTY if nan(x_max) else x_max]
        if i < self.N:
            d2 = 2 * (self.grid[i + 1] - self.grid[i])
            A += [[-d2, -1.0], [d2, 1.0]]
            h += [ECOS_INFTY if nan(x_next_min) else -x_next_min, ECOS_INFTY if nan(x_next_max) else x_next_max]
        xb = self.xbound(i)
        if xb is not None:
            A += [[0.0, 1.0], [0.0, -1.0]]
            h += [min(ECOS_MAXX, xb[1]), -xb[0]]
        return np.array(A), np.array(h)

    def solve(self, i, g, x_min, x_max, x_next_min, x_next_max):
        """ecosWrapper.solve_stagewise_optim(i, None, g, ...): minimise g.(u, x).  (u, x) or None."""
        a, b, c = self.rows(i)
        A, h = self.linear(i, x_min, x_max, x_next_min, x_next_max)
        return solve_stage(a, b, c, self.ell, A, h, -np.asarray(g, dtype=np.float64))

    # the three passes, one stage at a time, fed with the *solver under test*'s neighbouring values
    def controllable_stage(self, i, K_next):
        """_one_step (reachability_algorithm.py:200-238): [K_lo, K_hi] of stage i given K[i+1]."""
        nan = float("nan")
        hi = self.solve(i, [1e-9, -1.0], nan, nan, K_next[0], K_next[1])   # g_upper (:229-233)
        lo = self.solve(i, [-1e-9, 1.0], nan, nan, K_next[0], K_next[1])   # -g_upper (:234-236)
        if lo is None or hi is None:
            return None
        return max(lo[1], 0.0), hi[1]

    def feasible_stage(self, i):
        """compute_feasible_sets (reachability_algorithm.py:131-164): all four bounds +-1e4."""
        lo = self.solve(i, [1e-9, 1.0], -FEAS_MAXX, FEAS_MAXX, -FEAS_MAXX, FEAS_MAXX)
        hi = self.solve(i, [-1e-9, -1.0], -FEAS_MAXX, FEAS_MAXX, -FEAS_MAXX, FEAS_MAXX)
        if lo is None or hi is None:
            return None
        return max(lo[1], 0.0), hi[1]

    def forward_stage(self, i, x, K_next):
        """TOPPRA._forward_step (time_optimal_algorithm.py:55-92): the greedy u at fixed x."""
        delta = self.grid[i + 1] - self.grid[i]
        z = self.solve(i, [-2 * delta, -1.0], x, x, K_next[0], K_next[1])
        return None if z is None else z[0]


def check_solution(problem, sol, stages, tol_x=1e-7, tol_u=None, feasible_sets=None):
    """Compare a solver's output for ONE trajectory with the independent method at the given stages.

    problem: RobustTrajectory; sol: dict(K [N+1][2], sd2 [N+1], u [N]) of an Ok trajectory.
    Every stage is checked on its own, fed with the solver's own K[i+1] / x_i, so deviations do not
    accumulate along the scan.  Returns dict of max deviations; raises AssertionError beyond tolerance."""
    K, sd2, u = sol["K"], sol["sd2"], sol["u"]
    dev = {"K": 0.0, "u": 0.0, "X": 0.0, "stages": 0}
    for i in stages:
        want = problem.controllable_stage(i, K[i + 1])
        assert want is not None, "independent solver: stage %d infeasible but the solver returned K" % i
        dK = max(abs(K[i, 0] - want[0]), abs(K[i, 1] - want[1]))
        assert dK <= tol_x, "K[%d] = %r, independent solver says %r" % (i, K[i], want)
        dev["K"] = max(dev["K"], dK)
        # forward step: the solver's u_i must be the largest feasible u at its own x_i (shrunk by the
        # reference's retry only when infeasible, which an Ok trajectory of these tests never needs)
        uw = problem.forward_stage(i, sd2[i], K[i + 1])
        if uw is not None and np.isfinite(u[i]):
            du = abs(u[i] - uw)
            lim = tol_u if tol_u is not None else 1e-6 * (1.0 + abs(uw))
            assert du <= lim, "u[%d] = %r, independent solver says %r" % (i, u[i], uw)
            dev["u"] = max(dev["u"], du)
        if feasible_sets is not None:
            xw = problem.feasible_stage(i)
            assert xw is not None
            dX = max(abs(feasible_sets[i, 0] - xw[0]), abs(feasible_sets[i, 1] - xw[1]))
            assert dX <= tol_x, "X[%d] = %r, independent solver says %r" % (i, feasible_sets[i], xw)
            dev["X"] = max(dev["X"], dX)
        dev["stages"] += 1
    return dev


def first_failed_stage(K):
    """Index of the stage at which a backward scan gave up (NaN row), or None."""
    bad = np.flatnonzero(np.isnan(K[:, 1]))
    return None if bad.size == 0 else int(bad.max())


def check_batch(data, ell, out, interpolation=True, stride=7, want_X=True, max_traj=None, tol_x=1e-7):
    """Run check_solution over the Ok trajectories of a batch result (dict of [B]-leading arrays) and
    confirm, for trajectories the solver gave up on, that the independent method finds the offending
    stage infeasible too.  Returns the aggregated deviations."""
    B = data["coef"].shape[0]
    agg = {"K": 0.0, "u": 0.0, "X": 0.0, "stages": 0, "failed_confirmed": 0}
    for b in range(B if max_traj is None else min(B, max_traj)):
        grid = data["grid"] if data["grid"].ndim == 1 else data["grid"][b]
        brk = data["breaks"] if data["breaks"].ndim == 1 else data["breaks"][b]
        vlim = None if data.get("vlim") is None else data["vlim"][b]
        P = RobustTrajectory(data["coef"][b], brk, grid, vlim, data["alim"][b], ell, interpolation)
        K = out["K"][b]
        if out["status"][b] == 0:
            sol = {k: out[k][b] for k in ("K", "sd2", "u")}
            X = out["X"][b] if (want_X and out.get("X") is not None) else None
            dev = check_solution(P, sol, range(b % stride, P.N, stride), tol_x=tol_x, feasible_sets=X)
            for k in ("K", "u", "X"):
                agg[k] = max(agg[k], dev[k])
            agg["stages"] += dev["stages"]
        else:
            i = first_failed_stage(K)
            if i is not None and i < P.N and not np.isnan(K[i + 1]).any():
                assert P.controllable_stage(i, K[i + 1]) is None, "stage %d of trajectory %d is solvable" % (i, b)
                agg["failed_confirmed"] += 1
    return agg

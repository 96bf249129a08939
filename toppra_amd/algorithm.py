"""TOPP-RA algorithm layer: the reference's ``toppra.algorithm.TOPPRA`` surface, running on the
MI355X one *pass* at a time, plus ``BatchTOPPRA`` for B trajectories per launch.

Reference: algorithm/algorithm.py:27-215 (ParameterizationData / ReturnCode / base class),
reachabilitybased/reachability_algorithm.py:14-431, time_optimal_algorithm.py:8-92.
Same constructor arguments, same return shapes, same exceptions for bad input, same
"failure is data" convention (NaN / None results + ``problem_data.return_code``).
"""
import enum
import logging
import time

import numpy as np

from . import batch as _batch
from . import exceptions
from . import interpolator as _interp
from .constants import SMALL
from .solverwrapper import hipDenseSeidelWrapper, hipRobustWrapper, hipSeidelWrapper

logger = logging.getLogger(__name__)


class ParameterizationReturnCode(enum.Enum):
    """Return codes of a parameterization attempt (algorithm.py:49-62)."""

    Ok = "Ok: Successful parametrization"
    ErrUnknown = "Error: Unknown issue"
    ErrShortPath = "Error: Input path is very short"
    FailUncontrollable = "Error: Instance is not controllable"
    ErrForwardPassFail = "Error: Forward pass fail. Numerical errors occured"

    def __str__(self):
        return super(ParameterizationReturnCode, self).__repr__()


_STATUS_TO_CODE = {0: ParameterizationReturnCode.Ok, 1: ParameterizationReturnCode.FailUncontrollable,
                   2: ParameterizationReturnCode.ErrUnknown}


class ParameterizationData(object):
    """Internal data and output (algorithm.py:27-46)."""

    def __init__(self):
        self.return_code = ParameterizationReturnCode.ErrUnknown
        self.gridpoints = None
        self.sd_vec = None
        self.sdd_vec = None
        self.K = None
        self.X = None

    def __repr__(self):
        return "ParameterizationData(return_code:={}, N={:d})".format(
            self.return_code, self.gridpoints.shape[0])


class ParameterizationAlgorithm(object):
    """Base class: gridpoint selection and validation (algorithm.py:65-125)."""

    def __init__(self, constraint_list, path, gridpoints=None, parametrizer=None,
                 gridpt_max_err_threshold=1e-3, gridpt_min_nb_points=100):
        self.constraints = constraint_list
        self.path = path
        self._problem_data = ParameterizationData()
        if gridpoints is None:
            gridpoints = _interp.propose_gridpoints(path, max_err_threshold=gridpt_max_err_threshold,
                                                    min_nb_points=gridpt_min_nb_points)
            logger.info("No gridpoint specified. Automatically choose a gridpoint with %d points",
                        len(gridpoints))
        if path.path_interval[0] != gridpoints[0] or path.path_interval[1] != gridpoints[-1]:
            raise ValueError("Invalid manually supplied gridpoints.")
        self.gridpoints = np.array(gridpoints)
        self._problem_data.gridpoints = np.array(gridpoints)
        self._N = len(gridpoints) - 1
        if np.any(np.diff(self.gridpoints) <= 0):
            logger.fatal("Input gridpoints are not monotonically increasing.")
            raise ValueError("Bad input gridpoints.")
        from . import parametrizer as tparam
        if parametrizer is None or parametrizer == "ParametrizeSpline":
            self.parametrizer = tparam.ParametrizeSpline
        elif parametrizer == "ParametrizeConstAccel":
            self.parametrizer = tparam.ParametrizeConstAccel
        else:
            self.parametrizer = parametrizer

    @property
    def problem_data(self):
        return self._problem_data

    def compute_parameterization(self, sd_start, sd_end, return_data=False):
        raise NotImplementedError

    def compute_trajectory(self, sd_start=0, sd_end=0):
        """Time-parameterized joint trajectory, or None when the path cannot be parameterized
        (algorithm.py:163-194)."""
        t0 = time.time()
        self.compute_parameterization(sd_start, sd_end)
        if self.problem_data.return_code != ParameterizationReturnCode.Ok:
            logger.warning("Fail to parametrize path. Return code: %s", self.problem_data.return_code)
            return None
        traj = self.parametrizer(self.path, self.problem_data.gridpoints, self.problem_data.sd_vec)
        logger.info("Finish parametrization in %.3f secs", time.time() - t0)
        return traj


class ReachabilityAlgorithm(ParameterizationAlgorithm):
    """Reachability-analysis algorithms over a solver wrapper
    (reachability_algorithm.py:14-431).  ``solver_wrapper`` may be None, "hip" or "seidel" -- all
    select the HIP seidel wrapper, the only solver of this build."""

    _SOLVERS = ("hip", "seidel")

    def __init__(self, constraint_list, path, gridpoints=None, solver_wrapper=None, parametrizer=None,
                 **kwargs):
        super(ReachabilityAlgorithm, self).__init__(constraint_list, path, gridpoints=gridpoints,
                                                    parametrizer=parametrizer, **kwargs)
        has_conic = any(getattr(c.get_constraint_type(), "value", None) == 1 for c in constraint_list)
        if solver_wrapper is None:
            solver_wrapper = "hip"
        if has_conic:
            # the reference needs ecos / cvxpy here (reachability_algorithm.py:78-84); this build
            # solves the same stage problems exactly on the GPU -- parity unpinned against ECOS, cross-checked at
            # 1e-7 against an independent exact solver
            assert solver_wrapper.lower() in ("hip", "ecos"), \
                "Problem has conic constraints, solver {:} is not suitable".format(solver_wrapper)
            self.solver_wrapper = hipRobustWrapper(self.constraints, self.path, self.gridpoints)
        else:
            assert solver_wrapper.lower() in self._SOLVERS, "Solver {:} not found".format(solver_wrapper)
            try:  # velocity + acceleration limits: rows regenerated on the GPU from the spline table (the fused kernels)
                self.solver_wrapper = hipSeidelWrapper(self.constraints, self.path, self.gridpoints,
                                                       solve_lp1d=True)
            except NotImplementedError:
                # any other canonical-linear list (second-order / torque constraints, the reference's own or hand-written
                # constraint objects): parameters from the constraints' callbacks, the scans on the dense-row entries
                self.solver_wrapper = hipDenseSeidelWrapper(self.constraints, self.path, self.gridpoints,
                                                            solve_lp1d=True)

    def compute_feasible_sets(self):
        """X[N+1, 2]: feasible squared velocities per gridpoint (NaN where infeasible)."""
        X = np.array(self.solver_wrapper.feasible_sets())
        self._problem_data.X = X
        return X

    def compute_reachable_sets(self, sdmin, sdmax):
        """L[N+1, 2]: squared velocities reachable from [sdmin^2, sdmax^2] at the start
        (reachability_algorithm.py:409-431; computes and stores the feasible sets on the way, like the
        reference).  A NaN row marks the stage that failed; the rows after it stay zero."""
        assert sdmin <= sdmax and 0 <= sdmin
        L, X = self.solver_wrapper.reachable_sets(sdmin, sdmax)
        self._problem_data.X = np.array(X)
        L = np.array(L)
        if np.isnan(L).any():
            i = int(np.argmax(np.isnan(L).any(axis=1)))
            logger.warning("L[{:d}]={:}. Path not parametrizable.".format(i, L[i]))
        return L

    def compute_controllable_sets(self, sdmin, sdmax):
        """K[N+1, 2]: controllable squared velocities; a NaN row marks the stage that failed and
        the rows above it stay zero, as in the reference."""
        assert sdmin <= sdmax and 0 <= sdmin
        K = np.array(self.solver_wrapper.controllable_sets(sdmin, sdmax))
        if np.isnan(K).any():
            i = int(np.argmax(np.isnan(K).any(axis=1)))
            logger.warning("A numerical error occurs: The controllable set at step "
                           "[{:d} / {:d}] can't be computed.".format(i, self._N + 1))
        return K

    def compute_parameterization(self, sd_start, sd_end, return_data=False):
        """Returns (sdd_vec[N], sd_vec[N+1], v_vec[N,0]) (+ K with return_data); Nones when the
        instance is not controllable (reachability_algorithm.py:240-376)."""
        if sd_end < 0 or sd_start < 0:
            raise exceptions.BadInputVelocities(
                "Negative path velocities: path velocities must be positive: (%s, %s)" % (sd_start, sd_end))
        out = self.solver_wrapper.parameterization(sd_start, sd_end)
        K = np.array(out["K"])
        status = int(out["status"])
        self._problem_data.return_code = _STATUS_TO_CODE[status]
        if status == 1:
            if not np.isnan(K).any():
                self._problem_data.K = K
                logger.warning("The initial velocity is not controllable. {:f} not in ({:f}, {:f})".format(
                    sd_start ** 2, K[0, 0], K[0, 1]))
            else:
                logger.warning("An error occurred when computing controllable velocities. "
                               "The path is not controllable, or is badly conditioned.")
            return (None, None, None, K) if return_data else (None, None, None)
        self._problem_data.K = K
        sd_vec = np.array(out["sd"])
        sdd_vec = np.array(out["u"])
        v_vec = np.zeros((self._N, 0))
        self._problem_data.sd_vec = sd_vec
        self._problem_data.sdd_vec = sdd_vec
        if return_data:
            return sdd_vec, sd_vec, v_vec, K
        return sdd_vec, sd_vec, v_vec


class TOPPRA(ReachabilityAlgorithm):
    """Time-optimal path parameterization by reachability analysis
    (time_optimal_algorithm.py:8-92).

    >>> inst = TOPPRA([pc_vel, pc_acc], path, gridpoints=ss)
    >>> sdd, sd, _ = inst.compute_parameterization(0, 0)
    """


class TOPPRAsd(ReachabilityAlgorithm):
    """TOPP-RA with a specified duration (desired_duration_algorithm.py:21-234): the fastest and the
    slowest parameterizations are computed and a convex combination with the desired duration is
    found by bisection.  Unachievable durations return the fastest / slowest one, as the reference."""

    def set_desired_duration(self, desired_duration):
        self.desired_duration = desired_duration

    def compute_parameterization(self, sd_start, sd_end, return_data=False, atol=1e-5):
        assert sd_end >= 0 and sd_start >= 0, "Path velocities must be positive"
        out = self.solver_wrapper.parameterization_sd(sd_start, sd_end, self.desired_duration, atol)
        K = np.array(out["K"])
        status = int(out["status"])
        self._problem_data.return_code = _STATUS_TO_CODE[status]
        if status == 1:
            if not np.isnan(K).any():
                self._problem_data.K = K
            return (None, None, None, K) if return_data else (None, None, None)
        self._problem_data.K = K
        sd_vec, sdd_vec = np.array(out["sd"]), np.array(out["u"])
        v_vec = np.zeros((self._N, 0))
        self._problem_data.sd_vec = sd_vec
        self._problem_data.sdd_vec = sdd_vec
        return (sdd_vec, sd_vec, v_vec, K) if return_data else (sdd_vec, sd_vec, v_vec)


class BatchTOPPRA(object):
    """B independent TOPP-RA problems of one shape solved in one launch.

    Parameters
    ----------
    coef, breaks : arrays [B, 4, nseg, d] and [nseg+1] (or [B, nseg+1]) -- cubic spline tables
        (``batch.spline_coefficients`` builds them from waypoints), numpy or torch-ROCm tensors.
    gridpoints : [N+1] shared or [B, N+1] per trajectory.
    vlim, alim : [B, d, 2] joint velocity / acceleration limits (either may be None).
    """

    def __init__(self, coef, breaks, gridpoints, vlim, alim, interpolation=True):
        self.coef, self.breaks, self.gridpoints = coef, breaks, gridpoints
        self.vlim, self.alim, self.interpolation = vlim, alim, interpolation

    @classmethod
    def from_waypoints(cls, knots, waypoints, gridpoints, vlim, alim, bc_type="not-a-knot", gpu_fit=True, **kw):
        """Build the batch from waypoints [B, m, d].  ``gpu_fit`` selects the batched GPU spline fit
        (``batch.spline_fit_batch``, bit-identical to scipy for the supported boundary conditions);
        otherwise one batched scipy ``CubicSpline`` call on the host."""
        if gpu_fit:
            coef, breaks = _batch.spline_fit_batch(knots, waypoints, bc_type)
        else:
            coef, breaks = _batch.spline_coefficients(knots, waypoints, bc_type)
        if not hasattr(gridpoints, "data_ptr"):
            gridpoints = np.asarray(gridpoints, dtype=np.float64)
        return cls(coef, breaks, gridpoints, vlim, alim, **kw)

    def compute_parameterization(self, sd_start=None, sd_end=None, want_sd=True, variant=0, want_K=True, want_u=True):
        """dict(sd2, sd, u, K, status): per-trajectory results; status 0/1/2 = Ok /
        FailUncontrollable / ErrUnknown, failed rows NaN-filled.  ``want_K`` / ``want_u`` = False leave the
        controllable sets / path accelerations in a device workspace (fewer bytes back to a host caller)."""
        return _batch.solve_batch(self.coef, self.breaks, self.gridpoints, self.vlim, self.alim,
                                  sd_start, sd_end, self.interpolation, want_sd=want_sd, variant=variant,
                                  want_K=want_K, want_u=want_u)

    def compute_parameterization_sd(self, desired_duration, sd_start=None, sd_end=None, atol=1e-5):
        """TOPPRAsd for the batch: dict(sd2, sd, u, K, status, alpha)."""
        return _batch.solve_desired_duration_batch(self.coef, self.breaks, self.gridpoints, self.vlim, self.alim,
                                                   desired_duration, sd_start, sd_end, atol)

    def compute_trajectory(self, sd_start=None, sd_end=None, parametrizer="ParametrizeSpline"):
        """``ParameterizationAlgorithm.compute_trajectory`` (algorithm/algorithm.py:174-215) for the batch:
        parameterize, then build the output trajectories q(t) with the reference's parametrizer --
        "ParametrizeSpline" (its default) or "ParametrizeConstAccel" -- entirely on the GPU.  Returns a
        :class:`BatchTrajectory`; trajectories that could not be parameterized have ``status != 0`` and
        NaN durations (the reference returns None for them)."""
        res = self.compute_parameterization(sd_start, sd_end, want_sd=True, want_K=False, want_u=False)  # retiming reads sd only
        if parametrizer == "ParametrizeSpline":
            sp = _batch.param_spline_batch(self.coef, self.breaks, self.gridpoints, res["sd"])
            return BatchTrajectory("spline", res, self, spline=sp)
        if parametrizer == "ParametrizeConstAccel":
            ts, us = _batch.const_accel_times_batch(self.gridpoints, res["sd"])
            return BatchTrajectory("const_accel", res, self, ts=ts, us=us)
        raise NotImplementedError("parametrizer %r (ParametrizeSpline and ParametrizeConstAccel are available)" % (parametrizer,))


    def compute_trajectory_samples(self, times, sd_start=None, sd_end=None, fractions=True, orders=(0,)):
        """``traj = compute_trajectory(); traj(ts, order)`` (examples/plot_kinematics.py:48-57) for the batch WITHOUT the
        spline's coefficient table in between (2.9 GB at 65536 x 7 x 200): parameterize, then q / dq/dt / d2q/dt2 of the
        reference's default parametrizer (ParametrizeSpline) at ``times`` -- [T] fractions of each trajectory's duration
        (``np.linspace(0, 1, T)``) or [B, T] (fractions, or absolute times with ``fractions=False``).  Returns
        dict(q / qd / qdd [B, T, d] for the requested orders, duration [B], status [B]); the same bits as
        ``compute_trajectory()`` followed by its evaluation.  Up to 16 dof."""
        res = self.compute_parameterization(sd_start, sd_end, want_sd=True, want_K=False, want_u=False)
        out = _batch.param_spline_sample_batch(self.coef, self.breaks, self.gridpoints, res["sd"], times, fractions=fractions,
                                               orders=orders)
        out["status"] = res["status"]
        return out

    def compute_controllable_sets(self, sdmin, sdmax):
        return _batch.controllable_sets_batch(self.coef, self.breaks, self.gridpoints, self.vlim,
                                              self.alim, sdmin, sdmax, self.interpolation)

    def compute_feasible_sets(self):
        return _batch.feasible_sets_batch(self.coef, self.breaks, self.gridpoints, self.vlim,
                                          self.alim, self.interpolation)

    def compute_reachable_sets(self, sdmin, sdmax):
        """L[B, N+1, 2] (reachability_algorithm.py:409-431 per trajectory)."""
        return _batch.reachable_sets_batch(self.coef, self.breaks, self.gridpoints, self.vlim, self.alim, sdmin, sdmax,
                                           self.interpolation)

    @staticmethod
    def return_codes(status):
        return [_STATUS_TO_CODE[int(s)] for s in np.asarray(status)]


class BatchTrajectory(object):
    """B output trajectories q_b(t) on the GPU (``AbstractGeometricPath`` surface, batched):
    ``duration`` [B], ``__call__(times [B, T], order) -> [B, T, d]``, ``status`` [B]."""

    def __init__(self, kind, result, problem, spline=None, ts=None, us=None):
        self.kind, self.result, self._p = kind, result, problem
        self.status = result["status"]
        self._sp, self._ts, self._us = spline, ts, us

    @property
    def duration(self):
        """[B] seconds (NaN where the parameterization failed)."""
        status = self.status
        if self.kind == "spline":
            tk, cnt = self._sp["knot_times"], self._sp["counts"]
            if hasattr(tk, "gather"):  # torch
                dur = tk.gather(1, (cnt.long() - 1).clamp(min=0)[:, None])[:, 0]
                return dur.masked_fill(status.to(dur.device) != 0, float("nan"))
            dur = tk[np.arange(tk.shape[0]), np.maximum(cnt - 1, 0)]
        else:
            dur = self._ts[:, -1]
            if hasattr(dur, "masked_fill"):
                return dur.masked_fill(status.to(dur.device) != 0, float("nan"))
        return np.where(np.asarray(status) != 0, np.nan, dur)

    def __call__(self, times, order=0):
        if self.kind == "spline":
            return _batch.ppoly_eval_batch(self._sp["coef"], self._sp["knot_times"], times, order, self._sp["counts"])
        return _batch.const_accel_eval_batch(self._p.coef, self._p.breaks, self._p.gridpoints, self.result["sd"],
                                             self._ts, self._us, times, order)

"""Output parametrizers: sd(s) -> q(t).  Host-side numpy (a "next" row of SURVEY.md section 8f;
it runs once per trajectory after the hot path).  Reference: toppra/parametrizer.py:23-196.
"""
import numpy as np

from .constants import TINY
from .exceptions import ToppraError
from .interpolator import AbstractGeometricPath, SplineInterpolator


class ParametrizeConstAccel(AbstractGeometricPath):
    """Piecewise-constant path acceleration between gridpoints (parametrizer.py:23-158)."""

    def __init__(self, path, gridpoints, velocities):
        self._path = path
        self._ss = np.array(gridpoints, dtype=float)
        self._velocities = np.array(velocities, dtype=float)
        assert self._ss.ndim == 1 and self._ss.shape[0] == self._velocities.shape[0]
        assert np.all(self._velocities >= 0)
        self._xs = self._velocities ** 2
        ds = np.diff(self._ss)
        self._us = 0.5 * (self._xs[1:] - self._xs[:-1]) / ds
        dts = 2 * ds / (self._velocities[:-1] + self._velocities[1:])
        ts = np.zeros_like(self._ss)
        for i, dt in enumerate(dts):  # sequential sum, the reference's rounding order
            ts[i + 1] = ts[i] + dt
        self._ts = ts

    @property
    def dof(self):
        return self._path.dof

    @property
    def path_interval(self):
        return np.array([self._ts[0], self._ts[-1]])

    @property
    def duration(self):
        return self._ts[-1] - self._ts[0]

    def _eval_params(self, ts):
        idx = np.searchsorted(self._ts, ts, side="right") - 1
        idx = np.minimum(idx, len(self._us) - 1)
        dt = ts - self._ts[idx]
        us = self._us[idx]
        vs = self._velocities[idx] + dt * us
        ss = self._ss[idx] + dt * self._velocities[idx] + 0.5 * dt ** 2 * us
        return ss, vs, us

    def __call__(self, ts, order=0):
        scalar = isinstance(ts, (int, float))
        ts = np.atleast_1d(np.asarray(ts, dtype=float))
        ss, vs, us = self._eval_params(ts)
        if order == 0:
            out = self._path(ss)
        elif order == 1:
            out = self._path(ss, 1) * vs[:, None]
        elif order == 2:
            out = self._path(ss, 2) * vs[:, None] ** 2 + self._path(ss, 1) * us[:, None]
        else:
            raise ToppraError("Order %s is not supported." % order)
        return out[0] if scalar else out


class ParametrizeSpline(SplineInterpolator):
    """Cubic spline in time through q(s_i) at the gridpoint time stamps, clamped to q'(s) sd at both ends
    (parametrizer.py:161-196).  Host mirror of ``tpr_param_spline_batch`` (which BatchTOPPRA uses)."""

    def __init__(self, path, gridpoints, velocities):
        ss = np.asarray(gridpoints, dtype=float)
        sd = np.asarray(velocities, dtype=float)
        sd_avg = 0.5 * (sd[:-1] + sd[1:])
        # a standing stretch counts 5 s; np.cumsum accumulates left to right, i.e. the reference's running sum
        with np.errstate(divide="ignore", invalid="ignore"):
            dt = np.where(sd_avg > TINY, np.diff(ss) / np.where(sd_avg > TINY, sd_avg, 1.0), 5.0)
        stamps = np.concatenate([[0.0], np.cumsum(dt)])
        keep = np.concatenate([[True], dt >= TINY])   # gridpoints reached in no time are dropped
        ends = path.path_interval
        super(ParametrizeSpline, self).__init__(
            stamps[keep], path(ss[keep]),
            ((1, path(ends[0], 1) * sd[0]), (1, path(ends[1], 1) * sd[-1])))

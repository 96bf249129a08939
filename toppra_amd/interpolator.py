"""Geometric paths: the subset of toppra/interpolator.py the hot path consumes.

``SplineInterpolator`` keeps the reference's constructor and call signature
(interpolator.py:360-430); the fit is scipy's ``CubicSpline`` on the host (a "next" row of
SURVEY.md section 8f), and the kernels read its coefficient tensor ``cspl.c`` / ``cspl.x``.
"""
import numpy as np
from scipy.interpolate import CubicSpline


class AbstractGeometricPath(object):
    """Interface of a geometric path (interpolator.py:125-192)."""

    def __call__(self, path_positions, order=0):
        raise NotImplementedError

    @property
    def dof(self):
        raise NotImplementedError

    @property
    def path_interval(self):
        raise NotImplementedError

    @property
    def waypoints(self):
        raise NotImplementedError


class SplineInterpolator(AbstractGeometricPath):
    """Cubic spline through waypoints; ``bc_type`` as scipy ('not-a-knot' default, 'clamped',
    'natural', 'periodic', or explicit derivative tuples)."""

    def __init__(self, ss_waypoints, waypoints, bc_type="not-a-knot"):
        self.ss_waypoints = np.array(ss_waypoints)
        self._q_waypoints = np.array(waypoints)
        assert self.ss_waypoints.shape[0] == self._q_waypoints.shape[0]
        if len(self.ss_waypoints) == 1:
            # a single waypoint: constant path, zero derivatives (interpolator.py:398-417)
            q0 = self._q_waypoints[0]
            self.cspl = lambda s: np.broadcast_to(q0, np.shape(s) + np.shape(q0)).copy() if np.ndim(s) else q0
            self.cspld = lambda s: np.zeros(np.shape(s) + np.shape(q0))
            self.cspldd = self.cspld
        else:
            self.cspl = CubicSpline(ss_waypoints, waypoints, bc_type=bc_type)
            self.cspld = self.cspl.derivative()
            self.cspldd = self.cspld.derivative()

    def __call__(self, path_positions, order=0):
        if order == 0:
            return self.cspl(path_positions)
        if order == 1:
            return self.cspld(path_positions)
        if order == 2:
            return self.cspldd(path_positions)
        raise ValueError("Invalid order %s" % order)

    @property
    def waypoints(self):
        return self.ss_waypoints, self._q_waypoints

    @property
    def duration(self):
        return self.ss_waypoints[-1] - self.ss_waypoints[0]

    @property
    def path_interval(self):
        return np.array([self.ss_waypoints[0], self.ss_waypoints[-1]])

    @property
    def dof(self):
        if np.isscalar(self._q_waypoints[0]):
            return 1
        return self._q_waypoints[0].shape[0]


def spline_tables(path):
    """(coef [4, nseg, d], breaks [nseg+1]) of a cubic-spline path -- ours or the reference's
    ``toppra.SplineInterpolator`` (both expose a scipy PPoly as ``.cspl``)."""
    cspl = getattr(path, "cspl", None)
    if cspl is None or not hasattr(cspl, "c") or not hasattr(cspl, "x"):
        raise NotImplementedError(
            "the HIP path needs a cubic-spline geometric path exposing .cspl (SplineInterpolator)")
    c = np.asarray(cspl.c, dtype=np.float64)
    if c.shape[0] != 4:
        raise NotImplementedError("only cubic splines are supported, got order %d" % (c.shape[0] - 1))
    if c.ndim == 2:
        c = c[:, :, None]
    return np.ascontiguousarray(c), np.ascontiguousarray(cspl.x, dtype=np.float64)


def propose_gridpoints(path, max_err_threshold=1e-4, max_iteration=100, max_seg_length=0.05,
                       min_nb_points=100):
    """Gridpoints that cover ``path`` well enough (the reference's rule, interpolator.py:49-122): a segment
    is halved while it is longer than ``max_seg_length`` or its estimated interpolation error
    ``0.5 max|q''(mid)| ds^2`` exceeds ``max_err_threshold``; afterwards every segment is halved until
    there are ``min_nb_points`` points.  One vectorised pass over all segments per refinement level (a
    single path evaluation per level); the grids are the reference's, value for value."""
    pts = np.array([path.path_interval[0], path.path_interval[1]], dtype=float)
    last_pass = 0
    for last_pass in range(max_iteration):
        lo, hi = pts[:-1], pts[1:]
        mid, seg = 0.5 * (lo + hi), hi - lo
        curv = np.abs(0.5 * np.reshape(path(mid, 2), (len(mid), -1)) * (seg ** 2)[:, None]).max(axis=1)
        split = (seg > max_seg_length) | (curv > max_err_threshold)
        if not split.any():
            break
        pts = np.sort(np.concatenate([pts, mid[split]]))
    while len(pts) < min_nb_points:
        pts = np.sort(np.concatenate([pts, 0.5 * (pts[:-1] + pts[1:])]))
    # the reference's verdict (:119-120): failure is "the refinement used its last allowed pass" -- also when that
    # very pass found nothing left to split
    if last_pass == max_iteration - 1:
        raise ValueError("Unable to find a good gridpoint for this path.")
    return [float(v) for v in pts]

"""Batched TOPP-RA entry points over the C-ABI (numpy host arrays or torch-ROCm device tensors).

These are the array-level calls underneath the drop-in classes in :mod:`toppra_amd.algorithm`:
one call = B independent trajectories, each solved exactly as the reference's
``TOPPRA(..., solver_wrapper="seidel")`` would (reachability_algorithm.py:166-376).

With numpy inputs the library stages host<->device copies itself; with torch CUDA tensors the
device pointers are passed through and the kernels run on torch's current stream.
"""
import ctypes as C

import numpy as np

from . import _capi

__all__ = ["solve_batch", "controllable_sets_batch", "feasible_sets_batch", "reachable_sets_batch",
           "constraint_params_batch", "make_synthetic_batch", "spline_coefficients",
           "spline_fit_batch", "solve_batch_timed", "const_accel_times_batch", "const_accel_eval_batch",
           "solve_desired_duration_batch", "robust_solve_batch", "param_spline_batch", "ppoly_eval_batch"]


def _stream_ptr(like):
    if _capi.is_torch_cuda(like):
        import torch
        return C.c_void_p(torch.cuda.current_stream(like.device).cuda_stream)
    return None


def _empty(like, shape, dtype="f64"):
    if _capi.is_torch_cuda(like):
        import torch
        return torch.empty(shape, device=like.device,
                           dtype=torch.float64 if dtype == "f64" else torch.int32)
    # host results: page-locked memory (through torch's caching pinned allocator) when it pays -- the
    # device-to-host copy of the outputs dominates a host-buffer call, and from pageable memory it runs
    # at a third of the PCIe rate
    nbytes = int(np.prod(shape)) * (8 if dtype == "f64" else 4)
    if nbytes >= (1 << 20):
        try:
            import torch
            if torch.cuda.is_available():
                t = torch.empty(tuple(int(v) for v in shape), pin_memory=True,
                                dtype=torch.float64 if dtype == "f64" else torch.int32)
                return t.numpy()
        except Exception:
            pass
    return np.empty(shape, dtype=np.float64 if dtype == "f64" else np.int32)


def _prepare(coef):
    if _capi.is_torch_cuda(coef):
        dev = coef.device.index
        if dev is None:
            import torch
            dev = torch.cuda.current_device()
        _capi.init(dev)
    else:
        _capi.init()


def solve_batch(coef, breaks, grid, vlim, alim, sd_start=None, sd_end=None, interpolation=True,
                want_sd=False, variant=0, strict=False, want_K=True, want_u=True, active=None, sound=False):
    """compute_parameterization for B trajectories.

    Returns dict(sd2[B,N+1], u[B,N], K[B,N+1,2], status[B] (+ sd[B,N+1] if want_sd)); failed
    trajectories are NaN-filled with status 1 (FailUncontrollable) or 2 (ErrUnknown).
    ``want_K=False`` leaves the controllable sets in a device workspace (half of the output bytes of a
    host-buffer call); ``want_u=False`` the path accelerations too -- retiming (``compute_trajectory``)
    needs neither.

    ``strict=True`` (TPR_STRICT_SEIDEL) runs every stage LP through the reference's full Seidel
    iteration instead of answering it from a certified optimal vertex (same bits, slower).

    ``sound=True`` (TPR_SOUND_CERTIFICATES): the throughput kernels certify a MOVED active pair only where the
    reference's own pivot sequence is predictable (include/toppra_hip.h); the small-batch kernel always does.

    ``active`` [B, 4] int32 (in/out): the warm-start state ``active_c_up[2], active_c_down[2]`` of the
    reference's wrapper object, for sequences of passes on one instance (``None`` = a fresh instance)."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, sd_start, sd_end, interpolation,
                                 variant, strict=strict, active=active, sound=sound)
    B, N = p.B, p.N
    out = {"sd2": _empty(coef, (B, N + 1)), "status": _empty(coef, (B,), "i32")}
    if want_u:
        out["u"] = _empty(coef, (B, N))
    if want_K:
        out["K"] = _empty(coef, (B, N + 1, 2))
    if want_sd:
        out["sd"] = _empty(coef, (B, N + 1))
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out.get("sd")), u=_capi.ptr(out.get("u")),
                         K=_capi.ptr(out.get("K")), status=_capi.ptr(out["status"]))
    _capi.check(_capi.load().tpr_solve_batch(C.byref(p), C.byref(r), _stream_ptr(coef)))
    return out


def solve_desired_duration_batch(coef, breaks, grid, vlim, alim, desired_duration, sd_start=None, sd_end=None,
                                 atol=1e-5, variant=0, interpolation=True, squared=False):
    """TOPPRAsd.compute_parameterization for B trajectories (desired_duration_algorithm.py:42-191).

    ``desired_duration``: scalar or [B] seconds.  Returns dict(sd2, sd, u, K, status, alpha): alpha is
    the blend between the fastest (1) and slowest (0) parameterizations found by bisection.
    ``variant``: 0 = auto (from 9216 trajectories up to 8 dof, 14336 .. 36864 at 9 .. 15 dof: the certified lane kernel runs the backward scan and both
    forward profiles in one launch; rows across lanes otherwise), 2 / 3 force one.  ``squared``: sd_start / sd_end already
    hold sd^2 (TPR_BOUNDARY_SQUARED)."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, sd_start, sd_end, interpolation, variant=variant, squared=squared)
    B, N = p.B, p.N
    desired = _capi.per_traj_vector("desired_duration", desired_duration, B, coef)
    out = {"sd2": _empty(coef, (B, N + 1)), "sd": _empty(coef, (B, N + 1)), "u": _empty(coef, (B, N)),
           "K": _empty(coef, (B, N + 1, 2)), "status": _empty(coef, (B,), "i32"), "alpha": _empty(coef, (B,))}
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out["sd"]), u=_capi.ptr(out["u"]),
                         K=_capi.ptr(out["K"]), status=_capi.ptr(out["status"]))
    _capi.check(_capi.load().tpr_solve_desired_duration_batch(C.byref(p), _capi.ptr(desired), float(atol), C.byref(r),
                                                              _capi.ptr(out["alpha"]), _stream_ptr(coef)))
    return out


def robust_solve_batch(coef, breaks, grid, vlim, alim, ellipsoid, sd_start=None, sd_end=None, interpolation=True,
                       want_X=False, variant=0):
    """Robust TOPP-RA for B trajectories (``RobustLinearConstraint`` on the acceleration limits with
    perturbation ellipsoid ``(ru, rx, rc)``; BASELINE config 4).  PARITY UNPINNED against ECOS (absent here; the
    reference holds no golden vectors for it), cross-checked at 1e-7 against an independent exact solver
    (tests/test_gpu_robust.py): the reference solves these second-order-cone stage
    problems with ECOS; this solves the same problems exactly (see csrc/tpr_robust.hip.inc).  Returns dict(sd2, sd, u, K, status[, X]).
    ``variant``: 0 = auto (rows across lanes up to 16 dof), 1 = the generic one-trajectory-per-lane kernel -- same bits."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, sd_start, sd_end, interpolation, variant=variant)
    B, N = p.B, p.N
    ell = np.ascontiguousarray(np.asarray(ellipsoid, dtype=np.float64).reshape(3))  # always a host array
    out = {"sd2": _empty(coef, (B, N + 1)), "sd": _empty(coef, (B, N + 1)), "u": _empty(coef, (B, N)),
           "K": _empty(coef, (B, N + 1, 2)), "status": _empty(coef, (B,), "i32")}
    if want_X:
        out["X"] = _empty(coef, (B, N + 1, 2))
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out["sd"]), u=_capi.ptr(out["u"]),
                         K=_capi.ptr(out["K"]), status=_capi.ptr(out["status"]))
    _capi.check(_capi.load().tpr_robust_solve_batch(C.byref(p), ell.ctypes.data, C.byref(r), _capi.ptr(out.get("X")),
                                                    _stream_ptr(coef)))
    return out


def solve_batch_timed(coef, breaks, grid, vlim, alim, out, reps, sd_start=None, sd_end=None,
                      interpolation=True, variant=0, strict=False, sound=False):
    """bench.py helper: `reps` launches between two hipEvents on torch's current stream.
    Returns average ms per launch.  `out` is a dict from a previous solve_batch (device)."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, sd_start, sd_end, interpolation,
                                 variant, strict=strict, sound=sound)
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out.get("sd")), u=_capi.ptr(out["u"]),
                         K=_capi.ptr(out["K"]), status=_capi.ptr(out["status"]))
    ms = C.c_float(0)
    _capi.check(_capi.load().tpr_solve_batch_timed(C.byref(p), C.byref(r), _stream_ptr(coef), int(reps),
                                                   C.byref(ms)))
    return float(ms.value)


def controllable_sets_batch(coef, breaks, grid, vlim, alim, sdmin, sdmax, interpolation=True, active=None, squared=False,
                            variant=0, strict=False, sound=False):
    """compute_controllable_sets(sdmin, sdmax) for B trajectories -> K[B,N+1,2] (``active``: see solve_batch;
    ``squared``: sdmin / sdmax already hold sd^2 -- TPR_BOUNDARY_SQUARED; ``variant`` / ``strict`` / ``sound``: as solve_batch)."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, None, None, interpolation, active=active, squared=squared,
                                 variant=variant, strict=strict, sound=sound)
    sdmin = _capi.per_traj_vector("sdmin", sdmin, p.B, coef)
    sdmax = _capi.per_traj_vector("sdmax", sdmax, p.B, coef)
    K = _empty(coef, (p.B, p.N + 1, 2))
    _capi.check(_capi.load().tpr_controllable_sets_batch(C.byref(p), _capi.ptr(sdmin), _capi.ptr(sdmax),
                                                         _capi.ptr(K), _stream_ptr(coef)))
    return K


def solve_dense_batch(a, b, c, low, high, deltas, sd_start=None, sd_end=None, want_sd=False, squared=False, active=None):
    """compute_parameterization on DENSE rows -- any canonical-linear constraint list, flattened as the reference's
    seidelWrapper flattens it (cy_seidel_solverwrapper.pyx:425-531; :func:`toppra_amd.solverwrapper.dense_rows` does it for
    constraint objects): a, b, c [B, N+1, nC], low, high [B, N+1, 2], deltas [N] or [B, N].  Returns the dict of
    :func:`solve_batch`.  Every stage LP runs the reference's full Seidel iteration: the reference's bits.  ``active``
    (int32 [B, 4], in / out; all dense entries): the wrapper object's warm-start state, for passes chained on one object."""
    _prepare(a)
    p, keep = _capi.make_dense_problem(a, b, c, low, high, deltas, sd_start, sd_end, squared=squared, active=active)
    out = {"sd2": _empty(a, (p.B, p.N + 1)), "u": _empty(a, (p.B, p.N)), "K": _empty(a, (p.B, p.N + 1, 2)),
           "status": _empty(a, (p.B,), "i32")}
    if want_sd:
        out["sd"] = _empty(a, (p.B, p.N + 1))
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out.get("sd")), u=_capi.ptr(out["u"]), K=_capi.ptr(out["K"]),
                         status=_capi.ptr(out["status"]))
    _capi.check(_capi.load().tpr_solve_dense_batch(C.byref(p), C.byref(r), _stream_ptr(a)))
    return out


def solve_desired_duration_dense_batch(a, b, c, low, high, deltas, desired_duration, sd_start=None, sd_end=None, atol=1e-5,
                                       active=None, squared=False):
    """TOPPRAsd.compute_parameterization on dense rows (see :func:`solve_dense_batch`, :func:`solve_desired_duration_batch`):
    dict(sd2, sd, u, K, status, alpha)."""
    _prepare(a)
    p, keep = _capi.make_dense_problem(a, b, c, low, high, deltas, sd_start, sd_end, active=active, squared=squared)
    B, N = p.B, p.N
    desired = _capi.per_traj_vector("desired_duration", desired_duration, B, a)
    out = {"sd2": _empty(a, (B, N + 1)), "sd": _empty(a, (B, N + 1)), "u": _empty(a, (B, N)),
           "K": _empty(a, (B, N + 1, 2)), "status": _empty(a, (B,), "i32"), "alpha": _empty(a, (B,))}
    r = _capi.tpr_result(sd2=_capi.ptr(out["sd2"]), sd=_capi.ptr(out["sd"]), u=_capi.ptr(out["u"]),
                         K=_capi.ptr(out["K"]), status=_capi.ptr(out["status"]))
    _capi.check(_capi.load().tpr_solve_desired_duration_dense_batch(C.byref(p), _capi.ptr(desired), float(atol), C.byref(r),
                                                                    _capi.ptr(out["alpha"]), _stream_ptr(a)))
    return out


def controllable_sets_dense_batch(a, b, c, low, high, deltas, sdmin, sdmax, squared=False, active=None):
    """compute_controllable_sets(sdmin, sdmax) on dense rows (see :func:`solve_dense_batch`) -> K [B, N+1, 2]."""
    _prepare(a)
    p, keep = _capi.make_dense_problem(a, b, c, low, high, deltas, squared=squared, active=active)
    sdmin = _capi.per_traj_vector("sdmin", sdmin, p.B, a)
    sdmax = _capi.per_traj_vector("sdmax", sdmax, p.B, a)
    K = _empty(a, (p.B, p.N + 1, 2))
    _capi.check(_capi.load().tpr_controllable_sets_dense_batch(C.byref(p), _capi.ptr(sdmin), _capi.ptr(sdmax), _capi.ptr(K),
                                                               _stream_ptr(a)))
    return K


def reachable_sets_dense_batch(a, b, c, low, high, deltas, sdmin, sdmax, want_X=False, active=None, squared=False):
    """compute_reachable_sets(sdmin, sdmax) on dense rows (see :func:`solve_dense_batch`) -> L [B, N+1, 2] (and the feasible
    sets X it computes on the way with ``want_X``)."""
    _prepare(a)
    p, keep = _capi.make_dense_problem(a, b, c, low, high, deltas, active=active, squared=squared)
    sdmin = _capi.per_traj_vector("sdmin", sdmin, p.B, a)
    sdmax = _capi.per_traj_vector("sdmax", sdmax, p.B, a)
    L = _empty(a, (p.B, p.N + 1, 2))
    X = _empty(a, (p.B, p.N + 1, 2)) if want_X else None
    _capi.check(_capi.load().tpr_reachable_sets_dense_batch(C.byref(p), _capi.ptr(sdmin), _capi.ptr(sdmax), _capi.ptr(L),
                                                            _capi.ptr(X), _stream_ptr(a)))
    return (L, X) if want_X else L


def feasible_sets_dense_batch(a, b, c, low, high, deltas, active=None):
    """compute_feasible_sets on dense rows (see :func:`solve_dense_batch`) -> X [B, N+1, 2]."""
    _prepare(a)
    p, keep = _capi.make_dense_problem(a, b, c, low, high, deltas, active=active)
    X = _empty(a, (p.B, p.N + 1, 2))
    _capi.check(_capi.load().tpr_feasible_sets_dense_batch(C.byref(p), _capi.ptr(X), _stream_ptr(a)))
    return X


def reachable_sets_batch(coef, breaks, grid, vlim, alim, sdmin, sdmax, interpolation=True, want_X=False, squared=False):
    """compute_reachable_sets(sdmin, sdmax) for B trajectories -> L[B,N+1,2] (and the feasible sets
    X[B,N+1,2] it computes on the way with ``want_X``)."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, None, None, interpolation, squared=squared)
    sdmin = _capi.per_traj_vector("sdmin", sdmin, p.B, coef)
    sdmax = _capi.per_traj_vector("sdmax", sdmax, p.B, coef)
    L = _empty(coef, (p.B, p.N + 1, 2))
    X = _empty(coef, (p.B, p.N + 1, 2)) if want_X else None
    _capi.check(_capi.load().tpr_reachable_sets_batch(C.byref(p), _capi.ptr(sdmin), _capi.ptr(sdmax), _capi.ptr(L),
                                                      _capi.ptr(X), _stream_ptr(coef)))
    return (L, X) if want_X else L


def feasible_sets_batch(coef, breaks, grid, vlim, alim, interpolation=True, active=None, variant=0, strict=False,
                        sound=False):
    """compute_feasible_sets for B trajectories -> X[B,N+1,2] (``active``: see solve_batch).

    ``variant``: 0 = auto (one trajectory per wave for a handful of trajectories or with ``active``; the certified lane
    kernel from 8192 trajectories up to 8 dof; rows across lanes otherwise), 2 / 3 / 4 force a kernel family.
    ``strict`` (TPR_STRICT_SEIDEL): the reference's full iteration for every LP; ``sound``: see solve_batch."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, None, None, interpolation, active=active, variant=variant,
                                 strict=strict, sound=sound)
    X = _empty(coef, (p.B, p.N + 1, 2))
    _capi.check(_capi.load().tpr_feasible_sets_batch(C.byref(p), _capi.ptr(X), _stream_ptr(coef)))
    return X


def constraint_params_batch(coef, breaks, grid, vlim, alim, interpolation=True):
    """compute_constraint_params + seidelWrapper row build for B trajectories.

    Returns dict(a,b,c [B,N+1,nC], low, high, xbound [B,N+1,2], qs, qss [B,N+1,d])."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, vlim, alim, None, None, interpolation)
    nC = 2 + ((4 if interpolation else 2) * p.d if alim is not None else 0)
    B, N, d = p.B, p.N, p.d
    out = {k: _empty(coef, (B, N + 1, nC)) for k in ("a", "b", "c")}
    out.update({k: _empty(coef, (B, N + 1, 2)) for k in ("low", "high", "xbound")})
    out.update({k: _empty(coef, (B, N + 1, d)) for k in ("qs", "qss")})
    _capi.check(_capi.load().tpr_constraint_params_batch(
        C.byref(p), *[_capi.ptr(out[k]) for k in ("a", "b", "c", "low", "high", "xbound", "qs", "qss")],
        _stream_ptr(coef)))
    return out


# --------------------------------------------------------------------------------------------
# host-side input preparation (outside the hot path)

def spline_coefficients(knots, waypoints, bc_type="not-a-knot"):
    """Batched cubic-spline fit on the host: waypoints [B, m, d] -> coef [B, 4, m-1, d].

    One scipy ``CubicSpline`` call over trailing axes; the coefficients are bitwise identical to
    the per-trajectory ``SplineInterpolator(knots, waypoints[b]).cspl.c`` of the reference
    (interpolator.py:419)."""
    from scipy.interpolate import CubicSpline
    way = np.asarray(waypoints, dtype=np.float64)
    B, m, d = way.shape
    cs = CubicSpline(np.asarray(knots, dtype=np.float64), way.transpose(1, 0, 2), bc_type=bc_type)
    # cs.c: [4, m-1, B, d] -> [B, 4, m-1, d]
    return np.ascontiguousarray(cs.c.transpose(2, 0, 1, 3)), np.asarray(cs.x, dtype=np.float64)


def const_accel_times_batch(grid, sd):
    """ParametrizeConstAccel._process_parametrization for B trajectories: sd [B, N+1] ->
    (ts [B, N+1], us [B, N])."""
    _prepare(sd)
    dev = _capi.is_torch_cuda(sd)
    conv = (lambda x: x.contiguous()) if dev else _capi.f64
    sd, grid = conv(sd), conv(grid)
    B, n1 = (int(v) for v in sd.shape)
    p = _capi.tpr_problem(B=B, d=1, nseg=1, N=n1 - 1,
                          flags=(_capi.DEVICE_PTRS if dev else 0) | (_capi.GRID_PER_TRAJ if grid.ndim == 2 else 0))
    p.grid = _capi.ptr(grid)
    ts, us = _empty(sd, (B, n1)), _empty(sd, (B, n1 - 1))
    _capi.check(_capi.load().tpr_const_accel_times_batch(C.byref(p), _capi.ptr(sd), _capi.ptr(ts), _capi.ptr(us),
                                                         _stream_ptr(sd)))
    return ts, us


def const_accel_eval_batch(coef, breaks, grid, sd, ts, us, times, order=0):
    """ParametrizeConstAccel.__call__(t, order) for B trajectories: times [B, T] -> [B, T, d]."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, None, None)
    dev = _capi.is_torch_cuda(coef)
    conv = (lambda x: x.contiguous()) if dev else _capi.f64
    sd, ts, us, times = conv(sd), conv(ts), conv(us), conv(times)
    T = int(times.shape[1])
    out = _empty(coef, (p.B, T, p.d))
    _capi.check(_capi.load().tpr_const_accel_eval_batch(C.byref(p), _capi.ptr(sd), _capi.ptr(ts), _capi.ptr(us), T,
                                                        _capi.ptr(times), int(order), _capi.ptr(out),
                                                        _stream_ptr(coef)))
    return out


def param_spline_batch(coef, breaks, grid, sd, variant=0):
    """ParametrizeSpline (the reference's default output parametrizer, parametrizer.py:161-196) for B
    trajectories: sd [B, N+1] -> dict(knot_times [B, N+1], counts [B], coef [B, 4, N, d]): the cubic spline in
    time through q(s_i) at the gridpoint times, clamped to q'(s) sd at both ends.  Entries of
    ``knot_times`` from ``counts[b]`` on are padding (gridpoints reached in no time are dropped, as in the
    reference).  Evaluate with :func:`ppoly_eval_batch`.  ``variant``: 0 auto; 1 the generic two-kernel path (any d);
    2 the single fused kernel in LAPACK dgtsv's elimination order (same bits as 1); 3 the knot-parallel kernel (d <= 16,
    knots in LDS, cyclic reduction: knot derivatives equal to rounding, q(t) within the row's 1e-10; the default where it
    fits).  Knot times and counts are the same bits in every variant."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, None, None, variant=variant)
    dev = _capi.is_torch_cuda(coef)
    if dev:
        _capi.check_tensor("sd", sd, coef)
        sd = sd.contiguous()
    else:
        sd = _capi.f64(sd)
    if tuple(sd.shape) != (p.B, p.N + 1):
        raise ValueError("sd must have shape [B, N+1] = [%d, %d]" % (p.B, p.N + 1))
    out = {"knot_times": _empty(coef, (p.B, p.N + 1)), "counts": _empty(coef, (p.B,), "i32"),
           "coef": _empty(coef, (p.B, 4, p.N, p.d))}
    _capi.check(_capi.load().tpr_param_spline_batch(C.byref(p), _capi.ptr(sd), _capi.ptr(out["knot_times"]),
                                                    _capi.ptr(out["counts"]), _capi.ptr(out["coef"]), _stream_ptr(coef)))
    return out


def param_spline_sample_batch(coef, breaks, grid, sd, times, fractions=True, orders=(0,)):
    """ParametrizeSpline + its evaluation in ONE launch, without the [B, 4, N, d] coefficient table in between: what
    ``traj = inst.compute_trajectory(); traj(ts, order)`` computes, for B trajectories.  ``times``: [T] (fractions of each
    trajectory's duration: ``linspace(0, 1, T)``) or [B, T] (fractions, or absolute times with ``fractions=False``).
    Returns dict(q / qd / qdd [B, T, d] for the requested ``orders``, duration [B]) -- the same bits as
    :func:`param_spline_batch` (variant 3) followed by :func:`ppoly_eval_batch`.  d <= 16, knots in LDS."""
    _prepare(coef)
    p, keep = _capi.make_problem(coef, breaks, grid, None, None)
    dev = _capi.is_torch_cuda(coef)
    if dev:
        _capi.check_tensor("sd", sd, coef)
        _capi.check_tensor("times", times, coef)
        sd, times = sd.contiguous(), times.contiguous()
    else:
        sd, times = _capi.f64(sd), _capi.f64(times)
    if tuple(sd.shape) != (p.B, p.N + 1):
        raise ValueError("sd must have shape [B, N+1] = [%d, %d]" % (p.B, p.N + 1))
    if times.ndim not in (1, 2) or (times.ndim == 2 and int(times.shape[0]) != p.B):
        raise ValueError("times must have shape [T] or [B, T]")
    if times.ndim == 1 and not fractions:
        raise ValueError("shared sample times must be fractions of each trajectory's duration")
    T = int(times.shape[-1])
    names = {0: "q", 1: "qd", 2: "qdd"}
    out = {names[o]: _empty(coef, (p.B, T, p.d)) for o in orders}
    out["duration"] = _empty(coef, (p.B,))
    _capi.check(_capi.load().tpr_param_spline_sample_batch(
        C.byref(p), _capi.ptr(sd), T, _capi.ptr(times), int(times.ndim == 2), int(bool(fractions)),
        _capi.ptr(out.get("q")), _capi.ptr(out.get("qd")), _capi.ptr(out.get("qdd")), _capi.ptr(out["duration"]), _stream_ptr(coef)))
    return out


def ppoly_eval_batch(coef, breaks, times, order=0, counts=None):
    """SplineInterpolator.__call__(t, order) for B piecewise cubics with their own breakpoints:
    coef [B, 4, nseg, d], breaks [B, nseg+1], times [B, T] -> [B, T, d] (order 0 / 1 / 2; extrapolation from
    the end segments like scipy's PPoly).  ``counts`` [B] int32: breakpoints in use per path."""
    _prepare(coef)
    dev = _capi.is_torch_cuda(coef)
    if dev:
        for name, t in (("coef", coef), ("breaks", breaks), ("times", times)):
            _capi.check_tensor(name, t, coef)
        coef, breaks, times = coef.contiguous(), breaks.contiguous(), times.contiguous()
        if counts is not None:
            import torch
            if not (hasattr(counts, "is_cuda") and counts.is_cuda) or counts.device != coef.device:
                raise ValueError("counts must live on %s like coef" % (coef.device,))
            if counts.dtype != torch.int32:  # the kernel reads raw int32: another integer type is converted, not reinterpreted
                counts = counts.to(torch.int32)
            counts = counts.contiguous()
    else:
        coef, breaks, times = _capi.f64(coef), _capi.f64(breaks), _capi.f64(times)
        if counts is not None:
            counts = np.ascontiguousarray(counts, dtype=np.int32)
    B, four, nseg, d = (int(v) for v in coef.shape)
    if four != 4 or tuple(breaks.shape) != (B, nseg + 1) or times.ndim != 2 or int(times.shape[0]) != B:
        raise ValueError("need coef [B, 4, nseg, d], breaks [B, nseg+1], times [B, T]")
    if counts is not None and tuple(counts.shape) != (B,):
        raise ValueError("counts must have shape [B]")
    T = int(times.shape[1])
    out = _empty(coef, (B, T, d))
    _capi.check(_capi.load().tpr_ppoly_eval_batch(B, nseg, d, _capi.ptr(coef), _capi.ptr(breaks), _capi.ptr(counts), T,
                                                  _capi.ptr(times), int(order), _capi.ptr(out), int(dev), _stream_ptr(coef)))
    return out


_BC_KINDS = {"not-a-knot": 0, "clamped": 1, "natural": 2}


def _bc(bc, like, B, d):
    """scipy-style boundary condition -> (kind, value array or None)."""
    if isinstance(bc, str):
        if bc not in _BC_KINDS:
            raise NotImplementedError("bc_type %r is not supported on the GPU (use the scipy host path)" % (bc,))
        return _BC_KINDS[bc], None
    order, value = bc
    if order not in (1, 2):
        raise ValueError("boundary derivative order must be 1 or 2")
    if _capi.is_torch_cuda(like):
        import torch
        value = torch.as_tensor(value, dtype=torch.float64, device=like.device).expand(B, d).contiguous()
    else:
        value = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float64), (B, d)))
    return int(order), value


def spline_fit_batch(knots, waypoints, bc_type="not-a-knot"):
    """Batched cubic-spline fit on the GPU: waypoints [B, m, d] -> (coef [B, 4, m-1, d], breaks).

    Same construction as scipy's ``CubicSpline`` (which ``SplineInterpolator`` wraps,
    interpolator.py:419), arithmetic order included; ``bc_type`` is 'not-a-knot', 'clamped',
    'natural' or a pair ``((order, value), (order, value))`` with order 1 or 2.  numpy in -> numpy out,
    torch-ROCm tensors in -> tensors out."""
    _prepare(waypoints)
    dev = _capi.is_torch_cuda(waypoints)
    conv = (lambda x: x.contiguous()) if dev else _capi.f64
    if dev:
        import torch
        knots = torch.as_tensor(knots, dtype=torch.float64, device=waypoints.device)
    way, knots = conv(waypoints), conv(knots)
    if way.ndim != 3:
        raise ValueError("waypoints must have shape [B, m, d]")
    B, m, d = (int(v) for v in way.shape)
    if int(knots.shape[-1]) != m:
        raise ValueError("knots must have one entry per waypoint")
    bc = (bc_type, bc_type) if isinstance(bc_type, str) else bc_type
    k0, v0 = _bc(bc[0], way, B, d)
    k1, v1 = _bc(bc[1], way, B, d)
    coef = _empty(way, (B, 4, m - 1, d))
    _capi.check(_capi.load().tpr_spline_fit_batch(
        B, m, d, _capi.ptr(knots), int(knots.ndim == 2), _capi.ptr(way), k0, k1, _capi.ptr(v0), _capi.ptr(v1),
        _capi.ptr(coef), int(dev), _stream_ptr(way)))
    return coef, knots


def make_synthetic_batch(B, d, N, seed=20240924, n_waypoints=5):
    """The benchmark's synthetic random-spline batch, following
    examples/plot_kinematics.py:22-33: N(0,1) waypoints on linspace(0,1,5), symmetric limits
    vlim = 10+20 U, alim = 10+2 U, uniform grid, rest-to-rest."""
    rng = np.random.default_rng(seed)
    way = rng.standard_normal((B, n_waypoints, d))
    vmax = 10 + 20 * rng.random((B, d))
    amax = 10 + 2 * rng.random((B, d))
    knots = np.linspace(0, 1, n_waypoints)
    coef, breaks = spline_coefficients(knots, way)
    return {
        "coef": coef, "breaks": breaks, "grid": np.linspace(0, 1, N + 1),
        "vlim": np.ascontiguousarray(np.stack([-vmax, vmax], axis=-1)),
        "alim": np.ascontiguousarray(np.stack([-amax, amax], axis=-1)),
        "waypoints": way, "knots": knots,
    }

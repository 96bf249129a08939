"""ctypes binding of libtoppra_hip.so (include/toppra_hip.h).

This is the only place the Python host layer touches native code.  There is no CPU fallback:
if the library is missing or no gfx950 device is visible every compute entry raises.
"""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOPPRA_HIP_LIB") or os.path.join(HERE, "libtoppra_hip.so")

MAX_DOF = 32       # generic kernel
MAX_DOF_FAST = 16  # rows-across-lanes kernels
HAS_VELOCITY = 1
HAS_ACCELERATION = 2
ACC_INTERPOLATION = 4
DEVICE_PTRS = 8
BREAKS_PER_TRAJ = 16
GRID_PER_TRAJ = 32
STRICT_SEIDEL = 128
BOUNDARY_SQUARED = 256
SOUND_CERTIFICATES = 512

STATUS_OK, STATUS_FAIL_UNCONTROLLABLE, STATUS_ERR_UNKNOWN = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class ToppraHipError(RuntimeError):
    """The native library is missing, no MI355X is visible, or an API call failed."""


class tpr_problem(C.Structure):
    _fields_ = [("B", C.c_int32), ("d", C.c_int32), ("nseg", C.c_int32), ("N", C.c_int32),
                ("flags", C.c_int32), ("variant", C.c_int32),
                ("coef", C.c_void_p), ("breaks", C.c_void_p), ("grid", C.c_void_p),
                ("vlim", C.c_void_p), ("alim", C.c_void_p),
                ("sd_start", C.c_void_p), ("sd_end", C.c_void_p), ("active", C.c_void_p)]


class tpr_result(C.Structure):
    _fields_ = [("sd2", C.c_void_p), ("sd", C.c_void_p), ("u", C.c_void_p), ("K", C.c_void_p),
                ("status", C.c_void_p)]


class tpr_dense_problem(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("nC", C.c_int32), ("flags", C.c_int32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p),
                ("low", C.c_void_p), ("high", C.c_void_p), ("deltas", C.c_void_p),
                ("sd_start", C.c_void_p), ("sd_end", C.c_void_p), ("active", C.c_void_p)]


EXPORTS = (
    "tpr_init", "tpr_device_count", "tpr_last_error", "tpr_version", "tpr_abi_sizes", "tpr_solve_batch",
    "tpr_controllable_sets_batch", "tpr_feasible_sets_batch", "tpr_constraint_params_batch",
    "tpr_solve_stagewise_batch", "tpr_lp1d_batch", "tpr_lp2d_batch", "tpr_solve_batch_timed",
    "tpr_spline_fit_batch", "tpr_const_accel_times_batch", "tpr_const_accel_eval_batch",
    "tpr_solve_desired_duration_batch", "tpr_robust_solve_batch", "tpr_param_spline_batch", "tpr_ppoly_eval_batch",
    "tpr_reachable_sets_batch", "tpr_solve_dense_batch", "tpr_controllable_sets_dense_batch", "tpr_feasible_sets_dense_batch",
    "tpr_solve_desired_duration_dense_batch", "tpr_reachable_sets_dense_batch", "tpr_param_spline_sample_batch",
)

_lib = None
_lock = threading.Lock()
_inited_device = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (soname
    libamdhip64.so.7, but requested by torch under the unversioned name), so if this library pulled
    in /opt/rocm's copy first, a later ``import torch`` would load a second runtime that sees no
    GPU.  Importing torch first makes the dynamic loader bind our NEEDED libamdhip64.so.7 to the
    copy torch already loaded.  Set TOPPRA_HIP_NO_TORCH=1 to skip (torch-free deployments)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("TOPPRA_HIP_NO_TORCH"):
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load():
    """dlopen the library and declare signatures (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ToppraHipError(
                "libtoppra_hip.so is not built: run `python -m toppra_amd.build` "
                "(there is no CPU fallback for the TOPP-RA path)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        L.tpr_init.restype = C.c_int
        L.tpr_init.argtypes = [C.c_int]
        L.tpr_device_count.restype = C.c_int
        L.tpr_last_error.restype = C.c_char_p
        L.tpr_version.restype = C.c_char_p
        L.tpr_abi_sizes.restype = C.c_int
        L.tpr_abi_sizes.argtypes = [C.POINTER(C.c_int32)] * 3
        sizes = [C.c_int32(0), C.c_int32(0), C.c_int32(0)]
        L.tpr_abi_sizes(*[C.byref(v) for v in sizes])
        mine = [C.sizeof(tpr_problem), C.sizeof(tpr_result), C.sizeof(tpr_dense_problem)]
        if [v.value for v in sizes] != mine:
            raise ToppraHipError("libtoppra_hip.so was built from another header: its tpr_problem / tpr_result / "
                                 "tpr_dense_problem take %s bytes, this binding declares %s" % ([v.value for v in sizes], mine))
        P, R, V = C.POINTER(tpr_problem), C.POINTER(tpr_result), C.c_void_p
        L.tpr_solve_batch.restype = C.c_int
        L.tpr_solve_batch.argtypes = [P, R, V]
        L.tpr_solve_desired_duration_batch.restype = C.c_int
        L.tpr_solve_desired_duration_batch.argtypes = [P, V, C.c_double, R, V, V]
        L.tpr_robust_solve_batch.restype = C.c_int
        L.tpr_robust_solve_batch.argtypes = [P, V, R, V, V]
        L.tpr_solve_batch_timed.restype = C.c_int
        L.tpr_solve_batch_timed.argtypes = [P, R, V, C.c_int, C.POINTER(C.c_float)]
        L.tpr_controllable_sets_batch.restype = C.c_int
        L.tpr_controllable_sets_batch.argtypes = [P, V, V, V, V]
        L.tpr_reachable_sets_batch.restype = C.c_int
        L.tpr_reachable_sets_batch.argtypes = [P, V, V, V, V, V]
        L.tpr_feasible_sets_batch.restype = C.c_int
        L.tpr_feasible_sets_batch.argtypes = [P, V, V]
        L.tpr_constraint_params_batch.restype = C.c_int
        L.tpr_constraint_params_batch.argtypes = [P, V, V, V, V, V, V, V, V, V]
        L.tpr_solve_stagewise_batch.restype = C.c_int
        L.tpr_solve_stagewise_batch.argtypes = [P, V, V, V, V, C.c_int, V, V]
        L.tpr_spline_fit_batch.restype = C.c_int
        L.tpr_spline_fit_batch.argtypes = [C.c_int, C.c_int, C.c_int, V, C.c_int, V, C.c_int, C.c_int, V, V, V,
                                           C.c_int, V]
        L.tpr_const_accel_times_batch.restype = C.c_int
        L.tpr_const_accel_times_batch.argtypes = [P, V, V, V, V]
        L.tpr_const_accel_eval_batch.restype = C.c_int
        L.tpr_const_accel_eval_batch.argtypes = [P, V, V, V, C.c_int, V, C.c_int, V, V]
        L.tpr_param_spline_sample_batch.restype = C.c_int
        L.tpr_param_spline_sample_batch.argtypes = [P, V, C.c_int, V, C.c_int, C.c_int, V, V, V, V, V]
        L.tpr_param_spline_batch.restype = C.c_int
        L.tpr_param_spline_batch.argtypes = [P, V, V, V, V, V]
        L.tpr_ppoly_eval_batch.restype = C.c_int
        L.tpr_ppoly_eval_batch.argtypes = [C.c_int, C.c_int, C.c_int, V, V, V, C.c_int, V, C.c_int, V, C.c_int, V]
        DP = C.POINTER(tpr_dense_problem)
        L.tpr_solve_dense_batch.restype = C.c_int
        L.tpr_solve_dense_batch.argtypes = [DP, R, V]
        L.tpr_solve_desired_duration_dense_batch.restype = C.c_int
        L.tpr_solve_desired_duration_dense_batch.argtypes = [DP, V, C.c_double, R, V, V]
        L.tpr_reachable_sets_dense_batch.restype = C.c_int
        L.tpr_reachable_sets_dense_batch.argtypes = [DP, V, V, V, V, V]
        L.tpr_controllable_sets_dense_batch.restype = C.c_int
        L.tpr_controllable_sets_dense_batch.argtypes = [DP, V, V, V, V]
        L.tpr_feasible_sets_dense_batch.restype = C.c_int
        L.tpr_feasible_sets_dense_batch.argtypes = [DP, V, V]
        L.tpr_lp1d_batch.restype = C.c_int
        L.tpr_lp1d_batch.argtypes = [C.c_int, C.c_int] + [V] * 10
        L.tpr_lp2d_batch.restype = C.c_int
        L.tpr_lp2d_batch.argtypes = [C.c_int, C.c_int] + [V] * 12
        _lib = L
        return _lib


def last_error():
    return load().tpr_last_error().decode()


def check(rc):
    if rc != 0:
        raise ToppraHipError("libtoppra_hip: %s (code %d)" % (last_error(), rc))


def init(device=None):
    """Make ``device`` the default device of the calling thread's host-pointer calls; raises ToppraHipError when no
    gfx950 device is usable.  The library never moves HIP's current device for its caller: every entry point scopes
    itself to the device its data lives on (the device of the pointers, or this default for host arrays) and puts
    the caller's device back.  Called on every entry -- it is one table look-up plus, the first time a device is
    seen, the gfx950 check.  ``device=None``: the device of the last call, or LOCAL_RANK on the first."""
    global _inited_device
    L = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _inited_device is None else _inited_device
    check(L.tpr_init(int(device)))
    _inited_device = int(device)
    return _inited_device


def device_count():
    return load().tpr_device_count()


# --------------------------------------------------------------------------------------------
# argument marshalling: numpy (host) or torch CUDA tensors (device, zero-copy)

def is_torch_cuda(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and x.is_cuda


def f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def ptr(x):
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


def _per_traj(name, arr, B, like, dev):
    """A per-trajectory vector ([B], or a scalar broadcast to it) as a contiguous fp64 array on the
    problem's device."""
    if dev:
        import torch
        if not (hasattr(arr, "is_cuda") and arr.is_cuda):
            arr = torch.as_tensor(arr, dtype=torch.float64, device=like.device)
        check_tensor(name, arr, like)
        if arr.ndim == 0:
            arr = arr.expand(B)
        if tuple(arr.shape) != (B,):
            raise ValueError("%s must be a scalar or have shape [B] = [%d], got %s" % (name, B, tuple(arr.shape)))
        return arr.contiguous()
    arr = np.asarray(arr, dtype=np.float64)
    if arr.ndim == 0:
        arr = np.broadcast_to(arr, (B,))
    if arr.shape != (B,):
        raise ValueError("%s must be a scalar or have shape [B] = [%d], got %s" % (name, B, arr.shape))
    return np.ascontiguousarray(arr)


def make_dense_problem(a, b, c, low, high, deltas, sd_start=None, sd_end=None, squared=False, keep=None, active=None):
    """Build a tpr_dense_problem from the arrays of the reference's seidelWrapper (all numpy or all torch-CUDA):
    a, b, c [B, N+1, nC] (a_arr, b_arr, c_arr: nC counts the two reserved x_next rows), low, high [B, N+1, 2],
    deltas [N] or [B, N], sd_start / sd_end scalars or [B], active [B, 4] int32 (the wrapper object's warm-start state,
    in / out).  Shapes and dtypes are validated here: the C-ABI takes raw pointers."""
    dev = is_torch_cuda(a)
    keep = keep if keep is not None else []
    if dev:
        def conv(name, x):
            if not (hasattr(x, "is_cuda") and x.is_cuda):
                raise ValueError("%s must be a CUDA tensor like a (mixing host and device arrays is not supported)" % name)
            check_tensor(name, x, a)
            return x.contiguous()
        check_tensor("a", a, a)
    else:
        def conv(name, x):
            return f64(x)
    a = conv("a", a)
    if a.ndim != 3:
        raise ValueError("a must have shape [B, N+1, nC]")
    B, N1, nC = (int(s) for s in a.shape)
    N = N1 - 1
    if N < 1:
        raise ValueError("dense rows need at least two gridpoints")
    if not 2 <= nC <= 122:
        raise ValueError("nC = %d rows per stage (incl. the two x_next rows) is outside 2..122" % nC)
    b, c = conv("b", b), conv("c", c)
    low, high = conv("low", low), conv("high", high)
    for name, arr, shape in (("b", b, (B, N1, nC)), ("c", c, (B, N1, nC)), ("low", low, (B, N1, 2)), ("high", high, (B, N1, 2))):
        if tuple(arr.shape) != shape:
            raise ValueError("%s must have shape %s, got %s" % (name, shape, tuple(arr.shape)))
    deltas = conv("deltas", deltas)
    if deltas.ndim == 1:
        if int(deltas.shape[0]) != N:
            raise ValueError("deltas must have N = %d entries" % N)
        deltas = (deltas.unsqueeze(0).expand(B, N) if dev else np.broadcast_to(deltas, (B, N)))
        deltas = deltas.contiguous() if dev else np.ascontiguousarray(deltas)
    if tuple(deltas.shape) != (B, N):
        raise ValueError("deltas must have shape [N] or [B, N] = [%d, %d], got %s" % (B, N, tuple(deltas.shape)))
    p = tpr_dense_problem(B=B, N=N, nC=nC, flags=(DEVICE_PTRS if dev else 0) | (BOUNDARY_SQUARED if squared else 0))
    keep += [a, b, c, low, high, deltas]
    p.a, p.b, p.c, p.low, p.high, p.deltas = ptr(a), ptr(b), ptr(c), ptr(low), ptr(high), ptr(deltas)
    for name, arr in (("sd_start", sd_start), ("sd_end", sd_end)):
        if arr is not None:
            arr = _per_traj(name, arr, B, a, dev)
            keep.append(arr)
            setattr(p, name, ptr(arr))
    if active is not None:  # [B, 4] int32 warm-start state of the reference's wrapper object, updated in place
        if dev:
            import torch
            if not (hasattr(active, "is_cuda") and active.is_cuda) or active.device != a.device or \
                    active.dtype != torch.int32 or tuple(active.shape) != (B, 4) or not active.is_contiguous():
                raise ValueError("active must be a contiguous int32 tensor [B, 4] on a's device")
        elif not (isinstance(active, np.ndarray) and active.dtype == np.int32 and active.shape == (B, 4)
                  and active.flags["C_CONTIGUOUS"]):
            raise ValueError("active must be a C-contiguous int32 array [B, 4] (it is updated in place)")
        keep.append(active)
        p.active = ptr(active)
    return p, keep


def check_tensor(name, t, like):
    """Device tensors are passed to the kernels as raw pointers: they must be fp64 and live on the
    same device as ``coef``."""
    import torch
    if t.dtype != torch.float64:
        raise ValueError("%s must be float64 (got %s): the kernels read raw fp64 pointers" % (name, t.dtype))
    if not t.is_cuda or t.device != like.device:
        raise ValueError("%s must live on %s like coef (got %s)" % (name, like.device, t.device))


def per_traj_vector(name, arr, B, like):
    """Public form of the per-trajectory check for the other batch entries (sdmin, sdmax, ...)."""
    return _per_traj(name, arr, B, like, is_torch_cuda(like))


def make_problem(coef, breaks, grid, vlim, alim, sd_start=None, sd_end=None, interpolation=True,
                 variant=0, keep=None, strict=False, active=None, squared=False, sound=False):
    """Build a tpr_problem from arrays (all numpy or all torch-CUDA).  `keep` collects the
    converted arrays so they outlive the call.  Shapes and dtypes are validated here -- the C-ABI
    takes raw pointers and sizes, so a short or mistyped array would be read out of bounds:
    coef [B,4,nseg,d]; breaks [nseg+1] or [B,nseg+1]; grid [N+1] or [B,N+1] (strictly increasing);
    vlim/alim [B,d,2]; sd_start/sd_end scalars or [B]; active [B,4] int32 (in/out).  Device tensors must be float64
    on coef's device."""
    dev = is_torch_cuda(coef)
    keep = keep if keep is not None else []
    if dev:
        def conv(name, x):
            if not (hasattr(x, "is_cuda") and x.is_cuda):
                raise ValueError("%s must be a CUDA tensor like coef (mixing host and device arrays is not supported)" % name)
            check_tensor(name, x, coef)
            return x.contiguous()
        check_tensor("coef", coef, coef)
        coef = coef.contiguous()
    else:
        def conv(name, x):
            return f64(x)
        coef = f64(coef)
    breaks = conv("breaks", breaks)
    grid = conv("grid", grid)
    if coef.ndim != 4 or coef.shape[1] != 4:
        raise ValueError("coef must have shape [B, 4, nseg, d]")
    B, _, nseg, d = (int(s) for s in coef.shape)
    if grid.ndim not in (1, 2) or (grid.ndim == 2 and int(grid.shape[0]) != B):
        raise ValueError("grid must have shape [N+1] or [B, N+1] with B = %d, got %s" % (B, tuple(grid.shape)))
    if breaks.ndim not in (1, 2) or (breaks.ndim == 2 and int(breaks.shape[0]) != B):
        raise ValueError("breaks must have shape [nseg+1] or [B, nseg+1] with B = %d, got %s" % (B, tuple(breaks.shape)))
    N = int(grid.shape[-1]) - 1
    if N < 1:
        raise ValueError("grid needs at least two gridpoints")
    if not dev and not np.all(np.diff(grid, axis=-1) > 0):  # device grids are the caller's responsibility
        raise ValueError("grid must be strictly increasing")
    flags = (DEVICE_PTRS if dev else 0) | (STRICT_SEIDEL if strict else 0) | (BOUNDARY_SQUARED if squared else 0) | \
        (SOUND_CERTIFICATES if sound else 0)
    if breaks.ndim == 2:
        flags |= BREAKS_PER_TRAJ
    if grid.ndim == 2:
        flags |= GRID_PER_TRAJ
    if int(breaks.shape[-1]) != nseg + 1:
        raise ValueError("breaks must have nseg+1 entries")
    p = tpr_problem(B=B, d=d, nseg=nseg, N=N, flags=0, variant=int(variant))
    keep += [coef, breaks, grid]
    p.coef, p.breaks, p.grid = ptr(coef), ptr(breaks), ptr(grid)
    for name, arr, flag in (("vlim", vlim, HAS_VELOCITY), ("alim", alim, HAS_ACCELERATION)):
        if arr is not None:
            arr = conv(name, arr)
            if tuple(arr.shape) != (B, d, 2):
                raise ValueError("%s must have shape [B, d, 2] = [%d, %d, 2], got %s" % (name, B, d, tuple(arr.shape)))
            keep.append(arr)
            setattr(p, name, ptr(arr))
            flags |= flag
    if alim is not None and interpolation:
        flags |= ACC_INTERPOLATION
    for name, arr in (("sd_start", sd_start), ("sd_end", sd_end)):
        if arr is not None:
            arr = _per_traj(name, arr, B, coef, dev)
            keep.append(arr)
            setattr(p, name, ptr(arr))
    if active is not None:  # [B, 4] int32 warm-start state of the reference's wrapper object, updated in place
        if dev:
            import torch
            if not (hasattr(active, "is_cuda") and active.is_cuda) or active.device != coef.device or \
                    active.dtype != torch.int32 or tuple(active.shape) != (B, 4) or not active.is_contiguous():
                raise ValueError("active must be a contiguous int32 tensor [B, 4] on coef's device")
        elif not (isinstance(active, np.ndarray) and active.dtype == np.int32 and active.shape == (B, 4)
                  and active.flags["C_CONTIGUOUS"]):
            raise ValueError("active must be a C-contiguous int32 array [B, 4] (it is updated in place)")
        keep.append(active)
        p.active = ptr(active)
    p.flags = flags
    return p, keep

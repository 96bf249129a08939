"""Constraint classes of the hot path: joint velocity and joint acceleration limits.

Same class names, constructor arguments and error behaviour as toppra/constraint/
(linear_joint_velocity.py:8-53, linear_joint_acceleration.py:8-104, constraint.py:10-103).
``compute_constraint_params`` returns the reference's 7-tuple ``(a, b, c, F, g, ubound, xbound)``
but the numbers come from the HIP library (``tpr_constraint_params_batch``), not from numpy.
"""
from enum import Enum

import numpy as np

from . import batch as _batch
from .interpolator import spline_tables


class ConstraintType(Enum):
    Unknown = -1
    CanonicalLinear = 0
    CanonicalConic = 1


class DiscretizationType(Enum):
    Collocation = 0
    Interpolation = 1


class Constraint(object):
    """Base class (constraint.py:34-103)."""

    constraint_type = ConstraintType.Unknown
    discretization_type = DiscretizationType.Collocation
    n_extra_vars = 0
    dof = -1
    _format_string = ""

    def __repr__(self):
        return "%s(\n    Type: %s\n    Discretization Scheme: %s\n%s)" % (
            self.__class__.__name__, self.constraint_type, self.discretization_type, self._format_string)

    def get_dof(self):
        return self.dof

    def get_no_extra_vars(self):
        return self.n_extra_vars

    def get_constraint_type(self):
        return self.constraint_type

    def get_discretization_type(self):
        return self.discretization_type

    def set_discretization_type(self, discretization_type):
        if discretization_type in (0, DiscretizationType.Collocation):
            self.discretization_type = DiscretizationType.Collocation
        elif discretization_type in (1, DiscretizationType.Interpolation):
            self.discretization_type = DiscretizationType.Interpolation
        elif getattr(discretization_type, "value", None) in (0, 1):  # the reference's own enum
            self.discretization_type = DiscretizationType(discretization_type.value)
        else:
            raise NotImplementedError("Discretization type: %s not implemented!" % (discretization_type,))

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        raise NotImplementedError


class LinearConstraint(Constraint):
    """Canonical linear constraint a u + b x + c in {F v <= g} (linear_constraint.py:9-81)."""

    def __init__(self):
        self.constraint_type = ConstraintType.CanonicalLinear
        self.discretization_type = DiscretizationType.Collocation
        self.n_extra_vars = 0
        self.dof = -1
        self.identical = False


def _limits(lim, what):
    lim = np.array(lim, dtype=float)
    if np.isnan(lim).any():
        raise ValueError("Bad %s given: %s" % (what, lim))
    if lim.ndim == 1:
        lim = np.vstack((-lim, lim)).T
    assert lim.shape[1] == 2, "Wrong input shape."
    return lim


def _check_dof(constraint, path):
    if path.dof != constraint.get_dof():
        raise ValueError("Wrong dimension: constraint dof ({:d}) not equal to path dof ({:d})".format(
            constraint.get_dof(), path.dof))


def _params_on_device(path, gridpoints, vlim=None, alim=None, interpolation=True):
    coef, breaks = spline_tables(path)
    return _batch.constraint_params_batch(
        coef[None], breaks, np.asarray(gridpoints, dtype=np.float64),
        None if vlim is None else vlim[None], None if alim is None else alim[None], interpolation)


class JointVelocityConstraint(LinearConstraint):
    """vlim[j, 0] <= qdot_j <= vlim[j, 1]; becomes a bound on x = sd^2 at every gridpoint."""

    def __init__(self, vlim):
        super(JointVelocityConstraint, self).__init__()
        self.vlim = _limits(vlim, "velocity")
        self.dof = self.vlim.shape[0]
        for i in range(self.dof):
            if self.vlim[i, 0] >= self.vlim[i, 1]:
                raise ValueError("Bad velocity limits: {:} (lower limit) > {:} (higher limit)".format(
                    self.vlim[i, 0], self.vlim[i, 1]))
        self._format_string = "    Velocity limit: \n" + "".join(
            "      J{:d}: {:}\n".format(i + 1, self.vlim[i]) for i in range(self.dof))

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        _check_dof(self, path)
        out = _params_on_device(path, gridpoints, vlim=self.vlim)
        return None, None, None, None, None, None, np.array(out["xbound"][0])


class JointAccelerationConstraint(LinearConstraint):
    """alim[j, 0] <= q'_j u + q''_j x <= alim[j, 1] (Interpolation scheme by default)."""

    def __init__(self, alim, discretization_scheme=DiscretizationType.Interpolation):
        super(JointAccelerationConstraint, self).__init__()
        self.alim = _limits(alim, "velocity")  # sic: the reference's message says "velocity" too
        self.dof = self.alim.shape[0]
        self.set_discretization_type(discretization_scheme)
        self._format_string = "    Acceleration limit: \n" + "".join(
            "      J{:d}: {:}\n".format(i + 1, self.alim[i]) for i in range(self.dof))
        self.identical = True

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        _check_dof(self, path)
        interp = self.discretization_type == DiscretizationType.Interpolation
        out = _params_on_device(path, gridpoints, alim=self.alim, interpolation=interp)
        d = self.dof
        eye = np.vstack([np.eye(d), -np.eye(d)])
        g1 = np.concatenate([self.alim[:, 1], -self.alim[:, 0]])
        if not interp:
            a, b = out["a"][0, :, 2:2 + d], out["b"][0, :, 2:2 + d]
            return a, b, np.zeros_like(a), eye, g1, None, None
        # wrapper rows are [ +a_i | -a_i | +a~ | -a~ ]; the constraint-level a is [ a_i | a~ ]
        pick = np.r_[2:2 + d, 2 + 2 * d:2 + 3 * d]
        a, b = out["a"][0][:, pick], out["b"][0][:, pick]
        F = np.zeros((4 * d, 2 * d))
        F[:2 * d, :d] = eye
        F[2 * d:, d:] = eye
        return a, b, np.zeros_like(a), F, np.concatenate([g1, g1]), None, None


class JointVelocityConstraintVarying(LinearConstraint):
    """Joint velocity limits that vary along the path: ``vlim_func(s) -> [dof, 2]`` (linear_joint_velocity.py:55-87).
    Only ``xbound`` is produced.  Evaluated on the host (a Python callback per gridpoint, as in the reference) with the
    reference's arithmetic: the running bounds sdmin / sdmax are C floats (``_CythonUtils.pyx:60-101`` -- every min / max
    is rounded to fp32 on assignment, the upper bound is squared in fp32), the quotients are doubles.  Lists holding it
    run on the dense-row entries (hipDenseSeidelWrapper)."""

    def __init__(self, vlim_func):
        super(JointVelocityConstraintVarying, self).__init__()
        self.dof = np.shape(vlim_func(0))[0]
        self.vlim_func = vlim_func
        self._format_string = "    Varying Velocity limit: \n"

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        _check_dof(self, path)
        gridpoints = np.asarray(gridpoints, dtype=float)
        qs = np.asarray(path(gridpoints, 1), dtype=float)
        xbound = np.zeros((len(gridpoints), 2))
        for i, s in enumerate(gridpoints):
            vlim = np.asarray(self.vlim_func(s), dtype=float)
            sdmin, sdmax = np.float32(-1e8), np.float32(1e8)  # MAXSD
            for k in range(self.dof):
                if qs[i, k] > 0:
                    hi, lo = vlim[k, 1] / qs[i, k], vlim[k, 0] / qs[i, k]
                elif qs[i, k] < 0:
                    hi, lo = vlim[k, 0] / qs[i, k], vlim[k, 1] / qs[i, k]
                else:
                    continue
                sdmax = np.float32(hi if hi < float(sdmax) else float(sdmax))
                sdmin = np.float32(lo if lo > float(sdmin) else float(sdmin))
            lower = float(sdmin) if float(sdmin) > 0.0 else 0.0
            xbound[i] = lower * lower, float(sdmax * sdmax)
        return None, None, None, None, None, None, xbound


def colloc_to_interpolate(a, b, c, F, g, xbound, ubound, gridpoints, identical=False):
    """First-order interpolation form of canonical-linear parameters (linear_constraint.py:84-192): the constraint of
    stage i is imposed at gridpoint i and, through x_{i+1} = x_i + 2 delta_i u_i, at gridpoint i + 1 -- the row blocks
    [a_i | a_{i+1} + 2 delta_i b_{i+1}], [b_i | b_{i+1}], [c_i | c_{i+1}] against blkdiag(F_i, F_{i+1}), [g_i | g_{i+1}];
    the last stage repeats itself.  ``identical``: one F (m x d) and g (m) for every gridpoint.  Host numpy, like the
    reference: these parameters come from user callbacks (inverse dynamics) and feed the dense-row entries
    (tpr_*_dense_batch)."""
    if a is None:
        return None, None, None, None, None, xbound, ubound
    a, b, c = (np.asarray(v, dtype=float) for v in (a, b, c))
    two_delta = 2 * np.diff(np.asarray(gridpoints, dtype=float)).reshape(-1, 1)
    nxt = lambda v: np.concatenate((v[1:], v[-1:]), axis=0)  # noqa: E731  (gridpoint i + 1; the last one repeats itself)
    a_next = np.concatenate((a[1:] + two_delta * b[1:], a[-1:]), axis=0)
    a2, b2, c2 = np.hstack((a, a_next)), np.hstack((b, nxt(b))), np.hstack((c, nxt(c)))
    F, g = np.asarray(F, dtype=float), np.asarray(g, dtype=float)
    if identical:
        m, d = F.shape
        F2 = np.zeros((2 * m, 2 * d))
        F2[:m, :d] = F
        F2[m:, d:] = F
        g2 = np.concatenate((g, g))
    else:
        n1, m, d = F.shape
        F2 = np.zeros((n1, 2 * m, 2 * d))
        F2[:, :m, :d] = F
        F2[:, m:, d:] = nxt(F)
        g2 = np.hstack((g, nxt(g)))
    return a2, b2, c2, F2, g2, xbound, ubound


def _second_order_coefficients(inv_dyn, q, qs, qss):
    """(a, b, c)[N+1, m] of  w = a(s) sdd + b(s) sd^2 + c(s)  from an inverse-dynamics callback by substitution:
    c = tau(q, 0, 0), a = tau(q, 0, q') - c, b = tau(q, q', q'') - c (linear_second_order.py:154-162)."""
    zero = np.zeros(q.shape[1])
    c = np.array([inv_dyn(q_i, zero, zero) for q_i in q], dtype=float)
    a = np.array([inv_dyn(q_i, zero, qs_i) for q_i, qs_i in zip(q, qs)], dtype=float) - c
    b = np.array([inv_dyn(q_i, qs_i, qss_i) for q_i, qs_i, qss_i in zip(q, qs, qss)], dtype=float) - c
    return a, b, c


class SecondOrderConstraint(LinearConstraint):
    """General second-order constraint  A(q) qdd + qd^T B(q) qd + C(q) = w,  F(q) w <= g(q)  given by an inverse-dynamics
    callback ``inv_dyn(q, qd, qdd) -> w`` and callbacks ``constraint_F(q)``, ``constraint_g(q)``
    (linear_second_order.py:11-173); ``custom_term(path, s)`` is added to c (joint friction).  The parameters are
    evaluated on the host, through the user's callbacks, as in the reference; the solve runs on the dense-row entries
    (hipDenseSeidelWrapper)."""

    def __init__(self, inv_dyn, constraint_F, constraint_g, dof, custom_term=None,
                 discretization_scheme=DiscretizationType.Interpolation):
        super(SecondOrderConstraint, self).__init__()
        self.set_discretization_type(discretization_scheme)
        self.inv_dyn = inv_dyn
        self.constraint_F = constraint_F
        self.constraint_g = constraint_g
        self.dof = dof
        self.custom_term = custom_term
        self._format_string = "    Kind: Generalized Second-order constraint\n    Dimension:\n        F in R^({:d}, {:d})\n".format(
            *np.shape(constraint_F(np.zeros(dof))))

    @classmethod
    def joint_torque_constraint(cls, inv_dyn, taulim, joint_friction, **kwargs):
        """Joint torque bounds taulim [dof, 2] with dry friction joint_friction [dof] (sign(q') * friction added to c)."""
        taulim = np.asarray(taulim, dtype=float)
        dof = taulim.shape[0]
        F = np.vstack((np.eye(dof), -np.eye(dof)))
        g = np.concatenate((taulim[:, 1], -taulim[:, 0]))
        return cls(inv_dyn, lambda _q: F, lambda _q: g, dof,
                   lambda path, s: np.sign(path(s, 1)) * joint_friction, **kwargs)

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        _check_dof(self, path)
        gridpoints = np.asarray(gridpoints, dtype=float)
        q = np.asarray(path(gridpoints))
        a, b, c = _second_order_coefficients(self.inv_dyn, q, np.asarray(path(gridpoints, 1)), np.asarray(path(gridpoints, 2)))
        F = np.array([self.constraint_F(q_i) for q_i in q], dtype=float)
        g = np.array([self.constraint_g(q_i) for q_i in q], dtype=float)
        if self.custom_term is not None:
            for i, s in enumerate(gridpoints):
                c[i] = c[i] + self.custom_term(path, s)
        if self.discretization_type == DiscretizationType.Collocation:
            return a, b, c, F, g, None, None
        return colloc_to_interpolate(a, b, c, F, g, None, None, gridpoints)


class JointTorqueConstraint(LinearConstraint):
    """Joint torque bounds  tau_lim[:, 0] <= inv_dyn(q, qd, qdd) + fs_coef * sign(qd) <= tau_lim[:, 1]
    (joint_torque.py:10-116): one F = [I; -I], g = [tau_max; -tau_min] for every gridpoint (``identical``)."""

    def __init__(self, inv_dyn, tau_lim, fs_coef, discretization_scheme=DiscretizationType.Collocation):
        super(JointTorqueConstraint, self).__init__()
        self.inv_dyn = inv_dyn
        self.tau_lim = np.array(tau_lim, dtype=float)
        assert self.tau_lim.ndim == 2 and self.tau_lim.shape[1] == 2, "Wrong input shape."
        self.fs_coef = np.array(fs_coef, dtype=float)
        self.dof = self.tau_lim.shape[0]
        self.set_discretization_type(discretization_scheme)
        self.identical = True
        self._format_string = "    Torque limit: \n" + "".join(
            "      J{:d}: {:}\n".format(i + 1, lim) for i, lim in enumerate(self.tau_lim))

    def compute_constraint_params(self, path, gridpoints, *args, **kwargs):
        _check_dof(self, path)
        gridpoints = np.asarray(gridpoints, dtype=float)
        qs = np.asarray(path(gridpoints, 1))
        a, b, c = _second_order_coefficients(self.inv_dyn, np.asarray(path(gridpoints)), qs, np.asarray(path(gridpoints, 2)))
        for k in range(self.dof):  # dry friction
            c[:, k] += self.fs_coef[k] * np.sign(qs[:, k])
        eye = np.eye(self.dof)
        F = np.vstack((eye, -eye))
        g = np.concatenate((self.tau_lim[:, 1], -self.tau_lim[:, 0]))
        if self.discretization_type == DiscretizationType.Collocation:
            return a, b, c, F, g, None, None
        return colloc_to_interpolate(a, b, c, F, g, None, None, gridpoints, identical=True)


class ConicConstraint(Constraint):
    """Base class of canonical conic constraints (conic_constraint.py:6-44)."""

    def __init__(self):
        self.constraint_type = ConstraintType.CanonicalConic
        self.discretization_type = DiscretizationType.Collocation
        self.n_extra_vars = 0
        self.dof = -1
        self._format_string = ""


class RobustLinearConstraint(ConicConstraint):
    """Robustified canonical linear constraint (conic_constraint.py:47-124):
    ``a u + b x + c + ||diag(ru, rx, rc) [u, x, 1]||_2 <= 0`` for every row of the base constraint.

    ``compute_constraint_params`` returns the reference's 6-tuple
    ``(a, b, c, P, ubound, xbound)``; the rows come from the HIP library.  The second-order-cone
    stage problems (ECOS in the reference) are solved exactly on the GPU by
    ``solverwrapper.hipRobustWrapper`` -- parity unpinned against ECOS, cross-checked at 1e-7 against an
    independent exact solver.
    """

    def __init__(self, cnst, ellipsoid_axes_lengths, discretization_scheme=DiscretizationType.Collocation):
        super(RobustLinearConstraint, self).__init__()
        self.dof = cnst.get_dof()
        assert getattr(cnst.get_constraint_type(), "value", None) == 0  # CanonicalLinear
        self.set_discretization_type(discretization_scheme)
        if np.any(np.r_[ellipsoid_axes_lengths] < 0):
            raise ValueError("Perturbation must be non-negative. Input {:}".format(ellipsoid_axes_lengths))
        self.base_constraint = cnst
        self.ellipsoid_axes_lengths = ellipsoid_axes_lengths
        self._format_string += "    Robust constraint generated from a canonical linear constraint\n"

    def compute_constraint_params(self, path, gridpoints):
        base = self.base_constraint
        base.set_discretization_type(self.discretization_type)  # the reference mutates the base too
        if not hasattr(base, "alim"):
            raise NotImplementedError("robustification of %s is outside the HIP path" % type(base).__name__)
        _check_dof(base, path)
        interp = self.discretization_type == DiscretizationType.Interpolation
        out = _params_on_device(path, gridpoints, alim=np.ascontiguousarray(base.alim, dtype=np.float64),
                                interpolation=interp)
        # F a, F b, F c - g are the wrapper's dense rows 2.. (cy_seidel_solverwrapper.pyx:490-499)
        a, b, c = (np.array(out[k][0][:, 2:]) for k in ("a", "b", "c"))
        rows = a.shape[1]
        P = np.zeros((len(gridpoints), rows + 2, 3, 3))
        P[:] = np.diag(self.ellipsoid_axes_lengths)
        return a, b, c, P, None, None

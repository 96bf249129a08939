import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def batch_fixtures():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "batch_*.npz")))


def dense_fixtures():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "dense_*.npz")) if "reuse" not in os.path.basename(p))


def torque_model(mass, grav, cori):
    """The inverse dynamics of the dense fixtures (tools/make_golden.py::torque_model: same expression, same bits)."""
    M = np.diag(mass)
    return lambda q, qd, qdd: M.dot(qdd) + cori * np.sin(q) * (1 + qd * qd) + grav * np.cos(q)


def dense_constraints(fx, b, mod):
    """The constraint list of trajectory b of a dense fixture, built from the classes of `mod` (toppra_amd.constraint)."""
    DT = mod.DiscretizationType(int(fx["scheme"]))
    inv_dyn = torque_model(fx["mass"][b], fx["grav"][b], fx["cori"][b])
    taulim = np.stack([-fx["taumax"][b], fx["taumax"][b]], axis=1)
    cons = []
    for kind in str(fx["kinds"]).split(","):
        if kind == "vel":
            cons.append(mod.JointVelocityConstraint(np.stack([-fx["vmax"][b], fx["vmax"][b]], axis=1)))
        elif kind == "acc":
            cons.append(mod.JointAccelerationConstraint(np.stack([-fx["amax"][b], fx["amax"][b]], axis=1), discretization_scheme=DT))
        elif kind == "torque":
            cons.append(mod.JointTorqueConstraint(inv_dyn, taulim, fx["fric"][b], discretization_scheme=DT))
        elif kind == "second":
            cons.append(mod.SecondOrderConstraint.joint_torque_constraint(inv_dyn, taulim, fx["fric"][b], discretization_scheme=DT))
    return cons


def fixture_problem(fx):
    """(coef, breaks, grid, vlim, alim, sd_start, sd_end, interpolation) of a batch fixture."""
    return (fx["coef"], fx["breaks"], fx["grid"], fx.get("vlim"), fx.get("alim"), fx["sd_start"],
            fx["sd_end"], bool(int(fx["interpolation"])))


def assert_same(got, want, what, atol=0.0):
    """Bit-exact by default (NaNs must coincide); atol > 0 relaxes to |diff| <= atol."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what + ": NaN pattern differs"
    if atol == 0.0:
        if not np.array_equal(got, want, equal_nan=True):
            dev = np.nanmax(np.abs(got - want))
            raise AssertionError("%s not bit-exact, max deviation %g" % (what, dev))
    else:
        m = np.isfinite(want)
        dev = np.max(np.abs(got[m] - want[m])) if m.any() else 0.0
        assert dev <= atol, "%s deviates by %g > %g" % (what, dev, atol)


def path_from_tables(coef, breaks):
    """A toppra_amd geometric path from a stored coefficient table (fixtures hold scipy's CubicSpline.c / .x, not the
    waypoints): a SplineInterpolator whose spline objects are scipy PPolys over exactly those coefficients."""
    from scipy.interpolate import PPoly

    import toppra_amd as ta
    coef, breaks = np.asarray(coef, dtype=np.float64), np.asarray(breaks, dtype=np.float64)
    pp = PPoly(coef, breaks)
    path = ta.SplineInterpolator(breaks, pp(breaks))
    path.cspl = pp
    path.cspld = pp.derivative()
    path.cspldd = path.cspld.derivative()
    return path


def need_reference_solver():
    """The reference's compiled solver (oracle/_ref) or a SKIP -- but a FAILURE where it is expected: TOPPRA_EXPECT_REF=1
    (tools/gpu_full.sh, any push from the build container) or the marker oracle/_ref/EXPECTED, which __graft_entry__.build()
    writes after building the modules and which travels with them.  A snapshot that lost the binaries must not turn the
    reference-solver tests into silent skips."""
    import pytest
    from oracle import build_ref, ref_solver_baseline as rb
    if rb.available():
        return rb
    msg = "oracle/_ref holds no compiled reference solver (built where /root/reference exists)"
    if os.environ.get("TOPPRA_EXPECT_REF") == "1" or os.path.exists(os.path.join(build_ref.OUT, "EXPECTED")):
        pytest.fail(msg + " -- and it is expected here (TOPPRA_EXPECT_REF / oracle/_ref/EXPECTED)")
    pytest.skip(msg)

"""Live pinning of the CPU oracle against the REAL reference (build container only: needs
/root/reference; skipped on the GPU box).  Fresh seeds, bit-exact."""
import numpy as np
import pytest

from tests.helpers import assert_same

pytestmark = pytest.mark.reference


def _problem(rng, d, nway=5):
    way = rng.standard_normal((nway, d))
    vmax = 10 + 20 * rng.random(d)
    amax = 10 + 2 * rng.random(d)
    return way, np.stack([-vmax, vmax], 1), np.stack([-amax, amax], 1)


@pytest.mark.parametrize("d,N,scheme,sd", [(7, 200, 1, (0, 0)), (6, 500, 1, (0, 0)), (3, 60, 0, (0.1, 0.05)),
                                           (2, 40, 1, (3.0, 0.0)), (5, 77, 1, (0, 0.2))])
def test_parameterization_matches_reference(reference, oracle, d, N, scheme, sd):
    import toppra.algorithm as algo
    import toppra.constraint as constraint
    rng = np.random.default_rng(d * 1000 + N)
    knots = np.linspace(0, 1, 5)
    grid = np.linspace(0, 1, N + 1)
    for _ in range(12):
        way, vl, al = _problem(rng, d)
        path = reference.SplineInterpolator(knots, way)
        cons = [constraint.JointVelocityConstraint(vl),
                constraint.JointAccelerationConstraint(al, discretization_scheme=scheme)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        sdd, sdv, _, K = inst.compute_parameterization(sd[0], sd[1], return_data=True)
        flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if scheme else 0)
        w = oracle.Wrapper(path.cspl.c, path.cspl.x, grid, vl, al, flags=flags)
        st, osdd, osd, oxs, oK = w.compute_parameterization(*sd)
        assert_same(oK, K, "K")
        if sdv is None:
            assert st == 1
        else:
            assert_same(osd, sdv, "sd")
            assert_same(osdd, sdd, "sdd")
        # wrapper internals: dense rows and the variable box
        sw = inst.solver_wrapper
        X = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_feasible_sets()
        w2 = oracle.Wrapper(path.cspl.c, path.cspl.x, grid, vl, al, flags=flags)
        assert_same(w2.compute_feasible_sets(), X, "X")
        assert sw.get_no_stages() == N


@pytest.mark.parametrize("d,N,nway,seed", [(6, 90, 9, 1), (7, 120, 6, 2), (9, 50, 7, 3), (14, 40, 5, 4), (3, 150, 4, 5)])
def test_irregular_problems_match_reference(reference, oracle, d, N, nway, seed):
    """Asymmetric limits incl. positive lower velocity limits, joints that stand still, non-uniform knots
    and grids, non-zero boundary velocities, up to 14 dof: the oracle against the live reference."""
    import toppra.algorithm as algo
    import toppra.constraint as constraint
    rng = np.random.default_rng(9000 + seed)
    knots = np.concatenate([[0.0], np.sort(rng.random(nway - 2)) * 0.9 + 0.05, [1.0]])
    grid = 0.6 * np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]]) + 0.4 * np.linspace(0, 1, N + 1)
    seen = set()
    for _ in range(10):
        way = rng.standard_normal((nway, d))
        still = rng.random(d) < 0.15
        way = np.where(still[None, :], way[:1, :], way)
        vhi = 5 + 25 * rng.random(d); vlo = -(5 + 25 * rng.random(d))
        vlo = np.where(rng.random(d) < 0.05, 0.05 * rng.random(d), vlo)
        ahi = 5 + 10 * rng.random(d); alo = -(5 + 10 * rng.random(d))
        vl, al = np.stack([vlo, vhi], 1), np.stack([alo, ahi], 1)
        sd0 = 0.3 * rng.random() if rng.random() < 0.5 else 0.0
        sd1 = 0.3 * rng.random() if rng.random() < 0.5 else 0.0
        path = reference.SplineInterpolator(knots, way)
        cons = [constraint.JointVelocityConstraint(vl), constraint.JointAccelerationConstraint(al)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        sdd, sdv, _, K = inst.compute_parameterization(sd0, sd1, return_data=True)
        w = oracle.Wrapper(path.cspl.c, path.cspl.x, grid, vl, al)
        st, osdd, osd, oxs, oK = w.compute_parameterization(sd0, sd1)
        assert_same(oK, K, "K")
        seen.add(st)
        if sdv is None:
            assert st == 1
        else:
            assert_same(osd, sdv, "sd")
            assert_same(osdd, sdd, "sdd")
    assert 0 in seen


def test_lp2d_matches_reference(reference, oracle):
    import toppra.solverwrapper.cy_seidel_solverwrapper as seidel
    for seed in range(300):
        rng = np.random.default_rng(seed + 5000)
        n = int(rng.integers(1, 60))
        v = rng.standard_normal(3)
        a, b = rng.standard_normal((2, n))
        c = -rng.random(n) if seed % 2 == 0 else rng.standard_normal(n)
        low, high = np.array([-0.5, -0.9]), np.array([0.5, 0.9])
        ac = rng.integers(-1, n + 1, size=2)
        want = seidel.solve_lp2d(v, a, b, c, low, high, ac.astype(int))
        got = oracle.lp2d(v, a, b, c, low, high, ac)
        assert got[0] == want[0]
        if want[0]:
            assert got[1] == want[1] and list(got[2]) == list(want[2]) and list(got[3]) == list(want[3])


def test_velocity_bound_is_fp32(reference, oracle):
    """The fp32 rounding of the velocity bound (SURVEY.md trap 1) is reproduced exactly."""
    from toppra._CythonUtils import _create_velocity_constraint
    rng = np.random.default_rng(3)
    qs = rng.standard_normal((500, 7))
    qs[::17] = 0.0
    vmax = 10 + 20 * rng.random(7)
    vlim = np.stack([-vmax, 0.5 * vmax + rng.random(7)], 1)
    vlim[3] = [0.5, 2.0]  # positive lower limit exercises the sdmin branch
    _, _, cc = _create_velocity_constraint(qs, vlim)
    want = np.stack([cc[:, 1], -cc[:, 0]], 1)
    got = np.array([oracle.velocity_xbound(q, vlim) for q in qs])
    assert_same(got, want, "xbound")
    # and it is NOT what fp64 arithmetic would give
    assert np.any(got[:, 1] != np.array([np.min(np.where(q > 0, vlim[:, 1] / q, vlim[:, 0] / q)) ** 2
                                         for q in np.where(qs == 0, 1e-30, qs)]))


@pytest.mark.parametrize("d,N,scheme,sds", [(7, 100, 1, (0.0, 0.0)), (5, 60, 1, (0.0, 0.3)), (3, 40, 0, (0.1, 0.4)), (6, 80, 1, (0.5, 0.5))])
def test_reachable_sets_match_reference(reference, oracle, d, N, scheme, sds):
    """compute_reachable_sets (reachability_algorithm.py:378-431) incl. its deltas[i-1] objective and the
    warm-start state carried over from the feasible-set pass it runs first."""
    import toppra.algorithm as algo
    import toppra.constraint as constraint
    rng = np.random.default_rng(77 * d + N)
    knots = np.linspace(0, 1, 5)
    grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]])
    grid = 0.5 * grid + 0.5 * np.linspace(0, 1, N + 1)   # non-uniform: deltas[i-1] != deltas[i]
    for _ in range(8):
        way, vl, al = _problem(rng, d)
        path = reference.SplineInterpolator(knots, way)
        cons = [constraint.JointVelocityConstraint(vl),
                constraint.JointAccelerationConstraint(al, discretization_scheme=scheme)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        L = inst.compute_reachable_sets(*sds)
        flags = oracle.FLAG_VEL | oracle.FLAG_ACC | (oracle.FLAG_INTERP if scheme else 0)
        w = oracle.Wrapper(path.cspl.c, path.cspl.x, grid, vl, al, flags=flags)
        oL, oX = w.compute_reachable_sets(*sds)
        assert_same(oL, L, "L")
        assert_same(oX, inst.problem_data.X, "X")


def test_host_glue_matches_reference(reference):
    """propose_gridpoints and ParametrizeSpline (vectorised host code here, loops in the reference) give the
    reference's grids, knot times and spline coefficients value for value."""
    import toppra.interpolator as ri
    import toppra.parametrizer as rp
    import toppra_amd as ta
    from toppra_amd.interpolator import propose_gridpoints
    rng = np.random.default_rng(0)
    for t in range(20):
        d, m = int(rng.integers(1, 8)), int(rng.integers(3, 9))
        way = rng.standard_normal((m, d)) * float(rng.choice([0.1, 1, 5]))
        p1 = ta.SplineInterpolator(np.linspace(0, 1, m), way)
        p2 = reference.SplineInterpolator(np.linspace(0, 1, m), way)
        kw = dict(max_err_threshold=float(rng.choice([1e-4, 1e-3, 1e-5])), max_seg_length=float(rng.choice([0.05, 0.02, 0.2])),
                  min_nb_points=int(rng.choice([100, 30, 250])))
        g1, g2 = propose_gridpoints(p1, **kw), ri.propose_gridpoints(p2, **kw)
        assert np.array_equal(np.array(g1), np.array(g2))
        sd = np.abs(rng.standard_normal(len(g1)))
        sd[0] = sd[-1] = 0
        if t % 3 == 0:
            sd[5:8] = 0  # a standing stretch (the 5 s rule)
        a, b = ta.ParametrizeSpline(p1, g1, sd), rp.ParametrizeSpline(p2, g2, sd)
        assert np.array_equal(a.cspl.x, b.cspl.x) and np.array_equal(a.cspl.c, b.cspl.c)


@pytest.mark.parametrize("scheme", [0, 1])
def test_general_constraint_lists_match_reference(reference, oracle, scheme):
    """Constraint lists beyond velocity + acceleration, live: the reference's TOPPRA (seidel) on [varying velocity limits,
    JointTorqueConstraint, SecondOrderConstraint] against (1) toppra_amd's mirror classes (host numpy through the same
    callbacks) -- same parameters --, (2) toppra_amd.solverwrapper.dense_rows -- the wrapper's row assembly --, (3) the
    oracle's DenseWrapper on those rows: K, sd, u, feasible sets bit for bit."""
    import toppra.algorithm as algo
    import toppra.constraint as rc
    import toppra_amd as ta
    from tests.helpers import torque_model
    from toppra_amd.solverwrapper import dense_rows
    rng = np.random.default_rng(40 + scheme)
    knots, grid = np.linspace(0, 1, 5), np.linspace(0, 1, 41)
    for trial in range(6):
        d = int(rng.integers(2, 7))
        way = rng.standard_normal((5, d))
        inv_dyn = torque_model(1 + rng.random(d), 0.5 * rng.standard_normal(d), 0.3 * rng.standard_normal(d))
        taulim = np.stack([-6 - 6 * rng.random(d), 6 + 6 * rng.random(d)], axis=1)
        fric = 0.1 * rng.random(d)
        base = np.stack([-10 - 5 * rng.random(d), 10 + 5 * rng.random(d)], axis=1)
        vfun = lambda s: base * (1 + 0.4 * np.sin(5 * s))  # noqa: E731
        rpath, mpath = reference.SplineInterpolator(knots, way), ta.SplineInterpolator(knots, way)
        lists = []
        for mod in (rc, ta.constraint):
            DT = mod.DiscretizationType(scheme)
            lists.append([mod.JointVelocityConstraintVarying(vfun), mod.JointTorqueConstraint(inv_dyn, taulim, fric, discretization_scheme=DT),
                          mod.SecondOrderConstraint.joint_torque_constraint(inv_dyn, 1.3 * taulim, fric, discretization_scheme=DT)])
        rrows, mrows = dense_rows(lists[0], rpath, grid), dense_rows(lists[1], mpath, grid)
        for k in ("a", "b", "c", "low", "high", "deltas"):
            assert_same(mrows[k], rrows[k], "rows %s (mirror classes vs the reference's)" % k)
        sd0, sd1 = (0.0, 0.0) if trial % 2 else (0.1, 0.05)
        inst = algo.TOPPRA(lists[0], rpath, gridpoints=grid, solver_wrapper="seidel")
        sdd, sdv, _, K = inst.compute_parameterization(sd0, sd1, return_data=True)
        X = algo.TOPPRA(lists[0], rpath, gridpoints=grid, solver_wrapper="seidel").compute_feasible_sets()
        w = oracle.DenseWrapper(mrows["a"], mrows["b"], mrows["c"], mrows["low"], mrows["high"], mrows["deltas"])
        st, osdd, osd, oxs, oK = w.compute_parameterization(sd0, sd1)
        assert_same(oK, K, "K")
        if sdv is None:
            assert st == 1
        else:
            assert st == 0
            assert_same(osd, sdv, "sd")
            assert_same(osdd, sdd, "u")
        w2 = oracle.DenseWrapper(mrows["a"], mrows["b"], mrows["c"], mrows["low"], mrows["high"], mrows["deltas"])
        assert_same(w2.compute_feasible_sets(), X, "X")

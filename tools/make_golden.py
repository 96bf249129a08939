#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (hungpham2511/toppra, seidel solver)
in the build container.  The reference cannot travel to the GPU box, so its outputs are committed
as small fixtures; this script is their provenance.

    python tools/make_golden.py        # needs /root/reference; builds oracle/_ref on demand

Every fixture stores the inputs in the C-ABI layout (coef [B,4,nseg,d], breaks, grid, vlim, alim,
sd_start, sd_end) and the reference's outputs.  Trajectories the reference declares
FailUncontrollable (it returns None) are stored NaN-filled with status 1.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import ref_loader  # noqa: E402

ta = ref_loader.load()
if ta is None:
    raise SystemExit("reference not available")
import toppra.algorithm as algo  # noqa: E402
import toppra.constraint as constraint  # noqa: E402
from toppra.algorithm.algorithm import ParameterizationReturnCode as RC  # noqa: E402

STATUS = {RC.Ok: 0, RC.FailUncontrollable: 1, RC.ErrUnknown: 2}


def solve_one(knots, way, grid, vl, al, sd0=0.0, sd1=0.0, scheme=1, want_feasible=False, bc="not-a-knot"):
    path = ta.SplineInterpolator(knots, way, bc_type=bc)
    cons = []
    if vl is not None:
        cons.append(constraint.JointVelocityConstraint(vl))
    if al is not None:
        cons.append(constraint.JointAccelerationConstraint(al, discretization_scheme=scheme))
    inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
    sdd, sd, _, K = inst.compute_parameterization(sd0, sd1, return_data=True)
    N = len(grid) - 1
    st = STATUS[inst.problem_data.return_code]
    if sd is None:
        sd = np.full(N + 1, np.nan)
        sdd = np.full(N, np.nan)
    rec = {"coef": np.asarray(path.cspl.c), "breaks": np.asarray(path.cspl.x), "K": K, "sd": sd,
           "sd2": sd ** 2 if st else None, "u": sdd, "status": st}
    if want_feasible:
        # fresh instance: the reference's warm-start state is per object
        inst2 = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        rec["X"] = inst2.compute_feasible_sets()
    return rec, inst, path, cons


def xs_from(inst_or_none, sd):
    return sd


def batch_fixture(name, B, d, N, seed, sd_mode="zero", scheme=1, feasible=False, with_vel=True,
                  vscale=1.0, ascale=1.0, wayscale=None):
    rng = np.random.default_rng(seed)
    way = rng.standard_normal((B, 5, d))
    if wayscale is not None:  # tiny motions: the reference's seidel path degrades here
        way = way * (10.0 ** rng.uniform(wayscale[0], wayscale[1], size=(B, 1, 1)))
    vmax = (10 + 20 * rng.random((B, d))) * vscale
    amax = (10 + 2 * rng.random((B, d))) * ascale
    knots = np.linspace(0, 1, 5)
    grid = np.linspace(0, 1, N + 1)
    sd0 = np.zeros(B)
    sd1 = np.zeros(B)
    if sd_mode == "random":
        sd0 = 0.15 * rng.random(B)
        sd1 = 0.15 * rng.random(B)
        sd0[::5] = 6.0  # uncontrollable starts
    recs = []
    for b in range(B):
        vl = np.stack([-vmax[b], vmax[b]], axis=1) if with_vel else None
        al = np.stack([-amax[b], amax[b]], axis=1)
        rec, inst, path, cons = solve_one(knots, way[b], grid, vl, al, sd0[b], sd1[b], scheme, feasible)
        recs.append(rec)
    out = {
        "coef": np.stack([r["coef"] for r in recs]), "breaks": recs[0]["breaks"], "grid": grid,
        "alim": np.stack([-amax, amax], axis=-1), "sd_start": sd0, "sd_end": sd1,
        "K": np.stack([r["K"] for r in recs]), "sd": np.stack([r["sd"] for r in recs]),
        "u": np.stack([r["u"] for r in recs]), "status": np.array([r["status"] for r in recs], dtype=np.int32),
        "interpolation": np.array(scheme),
    }
    if with_vel:
        out["vlim"] = np.stack([-vmax, vmax], axis=-1)
    if feasible:
        out["X"] = np.stack([r["X"] for r in recs])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "status counts", np.bincount(out["status"], minlength=3))


def example_fixture():
    """Config 1: examples/plot_kinematics.py (seed 9, 7 dof) on a forced N=100 grid and on the
    automatically proposed grid; plus constraint parameters and a sample of stagewise solves."""
    np.random.seed(9)
    N_samples, dof = 5, 7
    way_pts = np.random.randn(N_samples, dof)
    knots = np.linspace(0, 1, 5)
    vlim_ = 10 + np.random.rand(dof) * 20
    alim_ = 10 + np.random.rand(dof) * 2
    vl = np.vstack((-vlim_, vlim_)).T
    al = np.vstack((-alim_, alim_)).T
    out = {"way_pts": way_pts, "knots": knots, "vlim": vl[None], "alim": al[None]}
    for tag, grid in (("n100", np.linspace(0, 1, 101)), ("auto", None)):
        path = ta.SplineInterpolator(knots, way_pts)
        if grid is None:
            inst0 = algo.TOPPRA([constraint.JointVelocityConstraint(vl), constraint.JointAccelerationConstraint(al)],
                                path, solver_wrapper="seidel")
            grid = inst0.gridpoints
        rec, inst, path, cons = solve_one(knots, way_pts, grid, vl, al, want_feasible=True)
        out["coef"] = rec["coef"][None]
        out["breaks"] = rec["breaks"]
        for k in ("K", "sd", "u", "X"):
            out[tag + "_" + k] = rec[k]
        out[tag + "_grid"] = np.asarray(grid)
        out[tag + "_status"] = np.array(rec["status"])
        # constraint parameters (compute_constraint_params) and the wrapper's dense rows
        pv = cons[0].compute_constraint_params(path, grid)
        pa = cons[1].compute_constraint_params(path, grid)
        out[tag + "_xbound"] = pv[6]
        out[tag + "_acc_a"], out[tag + "_acc_b"], out[tag + "_acc_c"] = pa[0], pa[1], pa[2]
        out[tag + "_acc_F"], out[tag + "_acc_g"] = pa[3], pa[4]
        out[tag + "_qs"] = path(grid, 1)
        out[tag + "_qss"] = path(grid, 2)
        if tag == "n100":
            # output parametrizers of the reference on this profile (parametrizer.py)
            ca = ta.ParametrizeConstAccel(path, grid, rec["sd"])
            out["ca_ts"], out["ca_us"] = ca._ts, ca._us
            tt = np.linspace(0, ca.duration, 257)
            out["ca_times"] = tt
            for order in (0, 1, 2):
                out["ca_q%d" % order] = ca(tt, order)
            sp = ta.ParametrizeSpline(path, grid, rec["sd"])
            out["spl_duration"] = np.array(sp.duration)
            ts2 = np.linspace(0, sp.duration, 129)
            out["spl_times"] = ts2
            for order in (0, 1, 2):
                out["spl_q%d" % order] = sp(ts2, order)
            # robust constraint parameters (conic_constraint.py:95-124), both discretisations
            for scheme in (0, 1):
                rc = constraint.RobustLinearConstraint(constraint.JointAccelerationConstraint(al),
                                                       [1e-3, 5e-2, 9e-3], scheme)
                ra, rb, rcc, rP, _, _ = rc.compute_constraint_params(path, grid)
                out["robust%d_a" % scheme], out["robust%d_b" % scheme] = ra, rb
                out["robust%d_c" % scheme], out["robust%d_P" % scheme] = rcc, rP
            # stagewise solves on a fresh wrapper, the reference's own call sequence preserved
            from toppra.solverwrapper.cy_seidel_solverwrapper import seidelWrapper
            for lp1d in (0, 1):
                w = seidelWrapper(cons, path, grid, solve_lp1d=lp1d)
                rng = np.random.default_rng(100 + lp1d)
                q = []
                r = []
                for _ in range(64):
                    i = int(rng.integers(0, 101))
                    g = rng.standard_normal(2)
                    mode = int(rng.integers(0, 4))
                    x = 5 * rng.random()
                    xb = [np.nan, np.nan, np.nan, np.nan]
                    if mode == 1:
                        xb = [x, x, 0.0, 50 * rng.random()]
                    elif mode == 2:
                        xb = [0.0, 30 * rng.random(), np.nan, 40 * rng.random()]
                    elif mode == 3:
                        xb = [2.0, 1.0, 0.0, 1.0]  # inverted bounds -> infeasible
                    res = np.array(w.solve_stagewise_optim(i, None, g, *xb))
                    q.append([i, g[0], g[1]] + xb)
                    r.append(res)
                out["stagewise_q_lp1d%d" % lp1d] = np.array(q)
                out["stagewise_r_lp1d%d" % lp1d] = np.array(r)
    np.savez_compressed(os.path.join(OUT, "example_kinematics_seed9.npz"), **out)
    print("example_kinematics_seed9: N auto =", len(out["auto_grid"]) - 1, "status", out["n100_status"], out["auto_status"])


def hard_fixture():
    """Asymmetric limits (incl. strictly positive lower velocity limits), 10 waypoints on random
    non-uniform knots, a non-uniform grid, one joint that does not move, non-zero end velocities."""
    rng = np.random.default_rng(2024)
    B, d, N, m = 32, 6, 90, 10
    knots = np.concatenate([[0.0], np.sort(rng.random(m - 2)) * 2.5 + 0.1, [2.8]])
    grid = np.concatenate([[knots[0]], np.sort(rng.random(N - 1)) * (knots[-1] - knots[0]) + knots[0], [knots[-1]]])
    recs = {k: [] for k in ("coef", "K", "sd", "u", "status", "X", "vlim", "alim")}
    sd1 = np.where(np.arange(B) % 3 == 0, 0.05, 0.0)
    for b in range(B):
        way = rng.standard_normal((m, d))
        if b % 4 == 0:
            way[:, 2] = way[0, 2]                       # joint 2 stands still
        vl = np.stack([-(2 + 20 * rng.random(d)), 5 + 20 * rng.random(d)], 1)
        al = np.stack([-(3 + 10 * rng.random(d)), 8 + 4 * rng.random(d)], 1)
        if b % 5 == 1:
            way = np.cumsum(np.abs(way), axis=0)          # monotone joints ...
            vl[:, 0] = 0.01 * rng.random(d)               # ... allow a positive lower velocity limit
        rec, inst, path, cons = solve_one(knots, way, grid, vl, al, 0.0, sd1[b], 1, want_feasible=True)
        for k in ("coef", "K", "sd", "u", "status", "X"):
            recs[k].append(rec[k])
        recs["vlim"].append(vl); recs["alim"].append(al)
    np.savez_compressed(os.path.join(OUT, "batch_d6_N90_hard.npz"), coef=np.stack(recs["coef"]), breaks=knots, grid=grid,
                        vlim=np.stack(recs["vlim"]), alim=np.stack(recs["alim"]), sd_start=np.zeros(B), sd_end=sd1,
                        K=np.stack(recs["K"]), sd=np.stack(recs["sd"]), u=np.stack(recs["u"]), X=np.stack(recs["X"]),
                        status=np.array(recs["status"], dtype=np.int32), interpolation=np.array(1))
    print("batch_d6_N90_hard status counts", np.bincount(recs["status"], minlength=3))


def sd_fixture(name="sd_batch_d5_N80", B=24, d=5, N=80, seed=31):
    """TOPPRAsd (desired_duration_algorithm.py) on random problems: unachievably short, in-range and
    unachievably long desired durations, some with boundary velocities."""
    rng = np.random.default_rng(seed)
    way = rng.standard_normal((B, 5, d))
    vmax = 10 + 20 * rng.random((B, d)); amax = 10 + 2 * rng.random((B, d))
    knots = np.linspace(0, 1, 5); grid = np.linspace(0, 1, N + 1)
    sd0 = np.where(np.arange(B) % 4 == 1, 0.05, 0.0); sd1 = np.where(np.arange(B) % 4 == 2, 0.03, 0.0)
    sd0[5] = 9.0  # uncontrollable
    recs = {k: [] for k in ("coef", "K", "sd", "u", "status", "desired")}
    for b in range(B):
        vl = np.stack([-vmax[b], vmax[b]], 1); al = np.stack([-amax[b], amax[b]], 1)
        path = ta.SplineInterpolator(knots, way[b])
        cons = [constraint.JointVelocityConstraint(vl), constraint.JointAccelerationConstraint(al)]
        fast = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_trajectory()
        base = fast.duration if fast is not None else 1.0
        desired = float(base * [0.5, 1.2, 1.7, 3.0, 10.0, 1e4][b % 6])
        inst = algo.TOPPRAsd(cons, path, gridpoints=grid, solver_wrapper="seidel")
        inst.set_desired_duration(desired)
        sdd, sd, _, K = inst.compute_parameterization(sd0[b], sd1[b], return_data=True)
        st = STATUS[inst.problem_data.return_code]
        if sd is None:
            sd = np.full(N + 1, np.nan); sdd = np.full(N, np.nan)
        recs["coef"].append(np.asarray(path.cspl.c)); recs["K"].append(K); recs["sd"].append(sd)
        recs["u"].append(sdd); recs["status"].append(st); recs["desired"].append(desired)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), coef=np.stack(recs["coef"]), breaks=knots, grid=grid,
                        vlim=np.stack([-vmax, vmax], -1), alim=np.stack([-amax, amax], -1), sd_start=sd0, sd_end=sd1,
                        desired=np.array(recs["desired"]), K=np.stack(recs["K"]), sd=np.stack(recs["sd"]),
                        u=np.stack(recs["u"]), status=np.array(recs["status"], dtype=np.int32))
    print(name, "status counts", np.bincount(recs["status"], minlength=3))


def cpp_fixture():
    """cpp/tests/test_algorithm.cpp:25-58 scenario (2 dof, Collocation, 51 gridpoints) solved by
    the Python seidel path; the C++ test's own 8-digit vectors are restated in
    tests/golden/kat_vectors.json."""
    knots = [0, 1, 2, 3]
    way = np.array([[0, 0], [1, 3], [2, 4], [0, 0]], dtype=float)
    grid = np.linspace(0, 3, 51)
    vl = np.array([[-1.0, 1.0]] * 2)
    al = np.array([[-0.2, 0.2]] * 2)
    rec, inst, path, cons = solve_one(knots, way, grid, vl, al, scheme=0, want_feasible=True)
    np.savez_compressed(os.path.join(OUT, "cpp_scenario_collocation.npz"), coef=rec["coef"][None],
                        breaks=rec["breaks"], grid=grid, vlim=vl[None], alim=al[None], K=rec["K"],
                        sd=rec["sd"], u=rec["u"], X=rec["X"], status=np.array(rec["status"]))
    print("cpp_scenario_collocation status", rec["status"])


def lp_fixture():
    """Random 2-D / 1-D LPs answered by the reference's own solve_lp2d / solve_lp1d (the shape of
    tests/tests/lpsolvers/seidel/test_lp2d.py:74-115, which needs cvxpy to run there)."""
    import toppra.solverwrapper.cy_seidel_solverwrapper as seidel
    n, d = 200, 50
    V, A, Bm, Cm, LO, HI, AC = [], [], [], [], [], [], []
    RES, VAL, VAR, ACO = [], [], [], []
    for seed in range(n):
        rng = np.random.default_rng(seed)
        v = rng.standard_normal(3)
        a, b = rng.standard_normal((2, d))
        c = -rng.random(d) if seed % 2 == 0 else rng.standard_normal(d)
        low = np.array([-0.5, -0.9])
        high = np.array([0.5, 0.9])
        if seed % 7 == 3:
            high = np.array([0.5, -1.0])  # inverted box
        ac = rng.integers(-2, d + 2, size=2)
        res, val, var, aco = seidel.solve_lp2d(v, a, b, c, low, high, ac.astype(int))
        V.append(v); A.append(a); Bm.append(b); Cm.append(c); LO.append(low); HI.append(high); AC.append(ac)
        RES.append(res)
        VAL.append(val if res else np.nan)
        VAR.append(np.array(var) if res else [np.nan, np.nan])
        ACO.append(np.array(aco) if res else [0, 0])
    out = dict(v=np.array(V), a=np.array(A), b=np.array(Bm), c=np.array(Cm), low=np.array(LO), high=np.array(HI),
               active_in=np.array(AC, dtype=np.int32), result=np.array(RES, dtype=np.int32), optval=np.array(VAL),
               optvar=np.array(VAR), active_out=np.array(ACO, dtype=np.int32))
    # 1-D
    V1, A1, B1, LO1, HI1, R1, VAL1, VAR1, AC1 = [], [], [], [], [], [], [], [], []
    for seed in range(n):
        rng = np.random.default_rng(1000 + seed)
        v = rng.standard_normal(2)
        a = rng.standard_normal(30)
        a[rng.random(30) < 0.2] = 1e-11  # exercised: |a| <= TINY rows are ignored
        b = rng.standard_normal(30) - (1.0 if seed % 2 else 0.0)
        low, high = -3 * rng.random(), 3 * rng.random()
        res, val, var, ac = seidel.solve_lp1d(v, a, b, low, high)
        V1.append(v); A1.append(a); B1.append(b); LO1.append(low); HI1.append(high)
        R1.append(res); VAL1.append(val if res else np.nan); VAR1.append(var if res else np.nan); AC1.append(ac if res else 0)
    out.update(v1=np.array(V1), a1=np.array(A1), b1=np.array(B1), low1=np.array(LO1), high1=np.array(HI1),
               result1=np.array(R1, dtype=np.int32), optval1=np.array(VAL1), optvar1=np.array(VAR1),
               active1=np.array(AC1, dtype=np.int32))
    np.savez_compressed(os.path.join(OUT, "random_lps.npz"), **out)
    print("random_lps: feasible 2-D", int(np.sum(out["result"])), "of", n, "; 1-D", int(np.sum(out["result1"])))


def reachable_fixture(name="reach_d5_N60", B=24, d=5, N=60, seed=41, scheme=1):
    """compute_reachable_sets (reachability_algorithm.py:409-431) on a non-uniform grid, per-trajectory
    [sdmin, sdmax] incl. sdmin == sdmax (the 1-D LP path of the first stage) and starts too fast to be feasible."""
    rng = np.random.default_rng(seed)
    knots = np.linspace(0, 1, 5)
    grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]])
    grid = 0.5 * grid + 0.5 * np.linspace(0, 1, N + 1)
    recs = {k: [] for k in ("coef", "vlim", "alim", "sdmin", "sdmax", "L", "X")}
    for b in range(B):
        way = rng.standard_normal((5, d))
        vmax = 10 + 20 * rng.random(d); amax = 10 + 2 * rng.random(d)
        vl, al = np.stack([-vmax, vmax], 1), np.stack([-amax, amax], 1)
        mode = b % 4
        sdmin = 0.0 if mode in (0, 1) else 0.3 * rng.random()
        sdmax = sdmin if mode in (0, 2) else sdmin + 0.5 * rng.random()
        if b % 11 == 10:
            sdmin = sdmax = 50.0   # far above the velocity limit: the first stage is infeasible
        path = ta.SplineInterpolator(knots, way)
        cons = [constraint.JointVelocityConstraint(vl), constraint.JointAccelerationConstraint(al, discretization_scheme=scheme)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        L = inst.compute_reachable_sets(sdmin, sdmax)
        for k, v in (("coef", np.asarray(path.cspl.c)), ("vlim", vl), ("alim", al), ("sdmin", sdmin), ("sdmax", sdmax),
                     ("L", L), ("X", inst.problem_data.X)):
            recs[k].append(v)
    out = {k: np.array(v) for k, v in recs.items()}
    out.update(breaks=np.asarray(path.cspl.x), grid=grid, interpolation=np.array(scheme))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, ": trajectories with a NaN stage", int(np.isnan(out["L"]).any(axis=(1, 2)).sum()), "of", B)


def reuse_fixture(name="reuse_d7_N100", B=24, d=7, N=100, seed=51):
    """Passes chained on ONE reference instance, in the order of examples/plot_kinematics.py (:48 compute_trajectory,
    :72 compute_feasible_sets) and on: compute_parameterization -> compute_feasible_sets ->
    compute_controllable_sets -> compute_parameterization again.  The wrapper object's warm-start state
    (active_c_up / active_c_down, cy_seidel_solverwrapper.pyx:526-527, also written by the forward pass's 1-D path
    :646-649) carries over from pass to pass, so every pass after the first pivots from where the previous one left
    off.  The same passes on FRESH instances are stored beside them (X_fresh, K2_fresh): where they differ, the
    carried state decided the bits."""
    rng = np.random.default_rng(seed)
    knots = np.linspace(0, 1, 5)
    grid = np.concatenate([[0.0], np.sort(rng.random(N - 1)), [1.0]])
    grid = 0.6 * grid + 0.4 * np.linspace(0, 1, N + 1)
    keys = ("coef", "vlim", "alim", "sd_start", "sd_end", "sdmin", "sdmax", "scheme", "K1", "sd1", "u1", "status1", "X",
            "K2", "K3", "sd3", "u3", "status3", "X_fresh", "K2_fresh")
    recs = {k: [] for k in keys}
    for b in range(B):
        way = rng.standard_normal((5, d)) * (1.0 if b % 6 else 10.0 ** rng.uniform(-3, 0))
        vmax = 10 + 20 * rng.random(d); amax = 10 + 2 * rng.random(d)
        vl, al = np.stack([-vmax, vmax], 1), np.stack([-amax, amax], 1)
        scheme = 0 if b % 4 == 3 else 1
        sd0 = 0.0 if b % 3 else 0.1 * rng.random()
        sd1 = 0.0 if b % 2 else 0.2 * rng.random()
        sdmin = 0.0 if b % 5 else 0.1 * rng.random()
        sdmax = sdmin + (0.0 if b % 2 else 0.3 * rng.random())
        path = ta.SplineInterpolator(knots, way)
        mk = lambda: algo.TOPPRA([constraint.JointVelocityConstraint(vl),
                                  constraint.JointAccelerationConstraint(al, discretization_scheme=scheme)],
                                 path, gridpoints=grid, solver_wrapper="seidel")
        inst = mk()

        def param(i):
            sdd, sd, _, K = i.compute_parameterization(sd0, sd1, return_data=True)
            st = STATUS[i.problem_data.return_code]
            if sd is None:
                sd, sdd = np.full(N + 1, np.nan), np.full(N, np.nan)
            return K, sd, sdd, st

        K1, s1, u1, st1 = param(inst)
        X = inst.compute_feasible_sets()
        K2 = inst.compute_controllable_sets(sdmin, sdmax)
        K3, s3, u3, st3 = param(inst)
        Xf = mk().compute_feasible_sets()
        K2f = mk().compute_controllable_sets(sdmin, sdmax)
        for k, v in zip(keys, (np.asarray(path.cspl.c), vl, al, sd0, sd1, sdmin, sdmax, scheme, K1, s1, u1, st1, X, K2, K3, s3, u3,
                               st3, Xf, K2f)):
            recs[k].append(v)
    out = {k: np.array(v) for k, v in recs.items()}
    out.update(breaks=np.asarray(path.cspl.x), grid=grid)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    same = lambda a, b: np.array([np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)])
    print(name, ": carried state changes bits of X in", int((~same(out["X"], out["X_fresh"])).sum()), ", of K2 in",
          int((~same(out["K2"], out["K2_fresh"])).sum()), ", of the second parameterization in",
          int((~(same(out["sd3"], out["sd1"]) & same(out["K3"], out["K1"]) & same(out["u3"], out["u1"]))).sum()), "of", B)


def pow_fixture(name="pow_boundary_d3_N40", B=16, d=3, N=40, seed=61):
    """Boundary velocities whose square the reference gets from libm: `sd ** 2` on Python floats is pow(sd, 2.0),
    which is NOT sd * sd for ~0.08 % of the doubles (one ulp).  Every sd_start / sd_end here is such a value, passed
    as a Python float, and the controllable-set call gets such sdmin / sdmax: K[N], K's lower rows near the end, and
    xs[0] carry the difference."""
    rng = np.random.default_rng(seed)

    def odd_square(lo, hi):
        while True:
            x = float(rng.uniform(lo, hi))
            if x ** 2 != x * x:
                return x

    knots = np.linspace(0, 1, 5)
    grid = np.linspace(0, 1, N + 1)
    keys = ("coef", "vlim", "alim", "sd_start", "sd_end", "sdmin", "sdmax", "K", "sd", "u", "status", "Kc")
    recs = {k: [] for k in keys}
    for b in range(B):
        way = rng.standard_normal((5, d))
        vmax = 10 + 20 * rng.random(d); amax = 10 + 2 * rng.random(d)
        vl, al = np.stack([-vmax, vmax], 1), np.stack([-amax, amax], 1)
        sd0, sd1 = odd_square(0.01, 0.3), odd_square(0.01, 0.3)
        sdmin = odd_square(0.0, 0.1)
        sdmax = odd_square(sdmin, sdmin + 0.2)
        path = ta.SplineInterpolator(knots, way)
        cons = [constraint.JointVelocityConstraint(vl), constraint.JointAccelerationConstraint(al)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        sdd, sd, _, K = inst.compute_parameterization(sd0, sd1, return_data=True)
        st = STATUS[inst.problem_data.return_code]
        if sd is None:
            sd, sdd = np.full(N + 1, np.nan), np.full(N, np.nan)
        Kc = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_controllable_sets(sdmin, sdmax)
        for k, v in zip(keys, (np.asarray(path.cspl.c), vl, al, sd0, sd1, sdmin, sdmax, K, sd, sdd, st, Kc)):
            recs[k].append(v)
    out = {k: np.array(v) for k, v in recs.items()}
    out.update(breaks=np.asarray(path.cspl.x), grid=grid)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, ": ok", int((out["status"] == 0).sum()), "of", B, "; K[N] differs from sd_end * sd_end in",
          int((out["K"][:, N, 0] != out["sd_end"] * out["sd_end"]).sum()))


def torque_model(mass, grav, cori):
    """A small rigid-body-like inverse dynamics tau(q, qd, qdd) = M qdd + cori sin(q) (1 + qd^2) + grav cos(q), rebuilt
    from the three stored vectors by tests/helpers.py::torque_model (same expression, same bits)."""
    M = np.diag(mass)
    return lambda q, qd, qdd: M.dot(qdd) + cori * np.sin(q) * (1 + qd * qd) + grav * np.cos(q)


def dense_fixture(name, B, d, N, seed, kinds, scheme, sd_mode="zero"):
    """Constraint lists beyond velocity + acceleration, the ones the dense-row entries (tpr_*_dense_batch) serve:
    kinds is a tuple of "vel", "acc", "torque" (JointTorqueConstraint), "second" (SecondOrderConstraint.joint_torque_constraint).
    Stored: the model's numbers (so that the GPU tests rebuild the constraint objects with toppra_amd's own classes), the
    reference's K / sd / u / status / feasible sets / controllable sets, and the dense rows of the reference's constraint
    objects flattened by toppra_amd.solverwrapper.dense_rows (host numpy; the oracle run on them reproduces the
    reference's outputs bit for bit: tests/test_oracle_golden.py)."""
    from toppra_amd.solverwrapper import dense_rows
    rng = np.random.default_rng(seed)
    knots, grid = np.linspace(0, 1, 5), np.linspace(0, 1, N + 1)
    way = rng.standard_normal((B, 5, d))
    mass, grav, cori = 1.0 + rng.random((B, d)), 0.5 * rng.standard_normal((B, d)), 0.3 * rng.standard_normal((B, d))
    taumax = 6.0 + 6.0 * rng.random((B, d))
    fric = 0.1 * rng.random((B, d))
    vmax, amax = 10 + 20 * rng.random((B, d)), 10 + 2 * rng.random((B, d))
    sd0, sd1 = np.zeros(B), np.zeros(B)
    if sd_mode == "random":
        sd0, sd1 = 0.15 * rng.random(B), 0.15 * rng.random(B)
        sd0[::4] = 6.0  # uncontrollable starts
    DT = constraint.DiscretizationType(scheme)
    out = {k: [] for k in ("a", "b", "c", "low", "high", "K", "sd", "u", "status", "X", "Kc")}
    for b in range(B):
        path = ta.SplineInterpolator(knots, way[b])
        inv_dyn = torque_model(mass[b], grav[b], cori[b])
        taulim = np.stack([-taumax[b], taumax[b]], axis=1)
        cons = []
        for kind in kinds:
            if kind == "vel":
                cons.append(constraint.JointVelocityConstraint(np.stack([-vmax[b], vmax[b]], axis=1)))
            elif kind == "acc":
                cons.append(constraint.JointAccelerationConstraint(np.stack([-amax[b], amax[b]], axis=1), discretization_scheme=DT))
            elif kind == "torque":
                cons.append(constraint.JointTorqueConstraint(inv_dyn, taulim, fric[b], discretization_scheme=DT))
            elif kind == "second":
                cons.append(constraint.SecondOrderConstraint.joint_torque_constraint(inv_dyn, taulim, fric[b], discretization_scheme=DT))
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        sdd, sd, _, K = inst.compute_parameterization(sd0[b], sd1[b], return_data=True)
        st = STATUS[inst.problem_data.return_code]
        if sd is None:
            sd, sdd = np.full(N + 1, np.nan), np.full(N, np.nan)
        out["K"].append(K); out["sd"].append(sd); out["u"].append(sdd); out["status"].append(st)
        out["X"].append(algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_feasible_sets())
        out["Kc"].append(algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_controllable_sets(0.05, 0.4))
        out.setdefault("L", []).append(algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_reachable_sets(0.0, 0.3))
        out.setdefault("L_point", []).append(algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_reachable_sets(0.1, 0.1))
        rows = dense_rows(cons, path, grid)
        for k in ("a", "b", "c", "low", "high"):
            out[k].append(rows[k])
        # TOPPRAsd on the same constraint list: unachievably short, in-range and unachievably long desired durations
        fast = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel").compute_trajectory(sd0[b], sd1[b])
        desired = float((fast.duration if fast is not None else 1.0) * [0.5, 1.3, 2.5, 1e4][b % 4])
        inst_sd = algo.TOPPRAsd(cons, path, gridpoints=grid, solver_wrapper="seidel")
        inst_sd.set_desired_duration(desired)
        sdd2, sd2v, _, K2 = inst_sd.compute_parameterization(sd0[b], sd1[b], return_data=True)
        if sd2v is None:
            sd2v, sdd2 = np.full(N + 1, np.nan), np.full(N, np.nan)
        for k, val in (("sd_desired", desired), ("sd_K", K2), ("sd_sd", sd2v), ("sd_u", sdd2),
                       ("sd_status", STATUS[inst_sd.problem_data.return_code])):
            out.setdefault(k, []).append(val)
    rec = {k: np.stack(v) for k, v in out.items()}
    rec["sd_status"] = rec["sd_status"].astype(np.int32)
    rec["status"] = rec["status"].astype(np.int32)
    rec.update(deltas=np.diff(grid), grid=grid, knots=knots, way=way, mass=mass, grav=grav, cori=cori, taumax=taumax, fric=fric,
               vmax=vmax, amax=amax, sd_start=sd0, sd_end=sd1, scheme=np.array(scheme), kinds=np.array(",".join(kinds)),
               sdmin_c=np.array(0.05), sdmax_c=np.array(0.4))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "nC", rec["a"].shape[2], "status counts", np.bincount(rec["status"], minlength=3))


def parametrizer_fixture(name="param_batch_d6_N150", B=10, d=6, N=150, seed=81):
    """The reference's output parametrizers on a batch: for each trajectory the time-optimal sd profile (boundary velocities
    on some), then ParametrizeSpline (the default of compute_trajectory) and ParametrizeConstAccel sampled at 97 times,
    orders 0 / 1 / 2, with their durations -- the f2 row's parity data beyond the single example trajectory."""
    rng = np.random.default_rng(seed)
    knots, grid = np.linspace(0, 1, 5), np.linspace(0, 1, N + 1)
    way = rng.standard_normal((B, 5, d))
    vmax, amax = 10 + 20 * rng.random((B, d)), 10 + 2 * rng.random((B, d))
    sd0 = np.where(np.arange(B) % 3 == 1, 0.08, 0.0); sd1 = np.where(np.arange(B) % 3 == 2, 0.05, 0.0)
    out = {k: [] for k in ("coef", "sd", "spl_duration", "spl_times", "spl_q0", "spl_q1", "spl_q2", "ca_duration", "ca_times",
                           "ca_q0", "ca_q1", "ca_q2")}
    for b in range(B):
        path = ta.SplineInterpolator(knots, way[b])
        cons = [constraint.JointVelocityConstraint(np.stack([-vmax[b], vmax[b]], 1)),
                constraint.JointAccelerationConstraint(np.stack([-amax[b], amax[b]], 1))]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        _, sd, _ = inst.compute_parameterization(sd0[b], sd1[b])
        assert sd is not None
        out["coef"].append(np.asarray(path.cspl.c)); out["sd"].append(sd)
        for key, par in (("spl", ta.ParametrizeSpline(path, grid, sd)), ("ca", ta.ParametrizeConstAccel(path, grid, sd))):
            ts = np.linspace(0, par.duration, 97)
            out[key + "_duration"].append(par.duration); out[key + "_times"].append(ts)
            for order in (0, 1, 2):
                out[key + "_q%d" % order].append(par(ts, order))
    rec = {k: np.stack(v) for k, v in out.items()}
    rec.update(breaks=knots, grid=grid, vlim=np.stack([-vmax, vmax], -1), alim=np.stack([-amax, amax], -1), sd_start=sd0, sd_end=sd1)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "durations", np.round(rec["spl_duration"], 3))


def dense_reuse_fixture(name="dense_reuse_d5_N60", B=16, d=5, N=60, seed=75):
    """Passes chained on ONE reference instance with a torque constraint, in script order: compute_parameterization ->
    compute_feasible_sets -> compute_controllable_sets -> compute_reachable_sets -> compute_parameterization again.  The
    wrapper object's warm-start state (active_c_up / active_c_down, written by the 2-D LPs and by the 1-variable path of
    the forward scan) is carried from pass to pass; `differs` counts the trajectories where a fresh instance would have
    returned other bits for the second and later passes."""
    from toppra_amd.solverwrapper import dense_rows
    rng = np.random.default_rng(seed)
    knots, grid = np.linspace(0, 1, 5), np.linspace(0, 1, N + 1)
    way = rng.standard_normal((B, 5, d))
    mass, grav, cori = 1.0 + rng.random((B, d)), 0.5 * rng.standard_normal((B, d)), 0.3 * rng.standard_normal((B, d))
    taumax, fric = 6.0 + 6.0 * rng.random((B, d)), 0.1 * rng.random((B, d))
    vmax, amax = 10 + 20 * rng.random((B, d)), 10 + 2 * rng.random((B, d))
    sd0, sd1 = 0.1 * rng.random(B), 0.1 * rng.random(B)
    out = {}
    differs = 0
    for b in range(B):
        path = ta.SplineInterpolator(knots, way[b])
        inv_dyn = torque_model(mass[b], grav[b], cori[b])
        cons = [constraint.JointVelocityConstraint(np.stack([-vmax[b], vmax[b]], axis=1)),
                constraint.JointAccelerationConstraint(np.stack([-amax[b], amax[b]], axis=1)),
                constraint.JointTorqueConstraint(inv_dyn, np.stack([-taumax[b], taumax[b]], axis=1), fric[b],
                                                 discretization_scheme=constraint.DiscretizationType.Interpolation)]
        inst = algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        sdd, sd, _, K = inst.compute_parameterization(sd0[b], sd1[b], return_data=True)
        assert sd is not None
        X = inst.compute_feasible_sets()
        Kc = inst.compute_controllable_sets(0.05, 0.4)
        L = inst.compute_reachable_sets(0.0, 0.3)
        sdd2, sd2v, _, K2 = inst.compute_parameterization(sd0[b], sd1[b], return_data=True)
        fresh = lambda: algo.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")  # noqa: E731
        same = (np.array_equal(X, fresh().compute_feasible_sets(), equal_nan=True)
                and np.array_equal(Kc, fresh().compute_controllable_sets(0.05, 0.4), equal_nan=True)
                and np.array_equal(L, fresh().compute_reachable_sets(0.0, 0.3), equal_nan=True)
                and np.array_equal(K2, K, equal_nan=True) and np.array_equal(sd2v, sd, equal_nan=True))
        differs += not same
        rows = dense_rows(cons, path, grid)
        for k, val in (("a", rows["a"]), ("b", rows["b"]), ("c", rows["c"]), ("low", rows["low"]), ("high", rows["high"]), ("K", K),
                       ("sd", sd), ("u", sdd), ("X", X), ("Kc", Kc), ("L", L), ("K2", K2), ("sd2nd", sd2v), ("u2nd", sdd2)):
            out.setdefault(k, []).append(val)
    rec = {k: np.stack(v) for k, v in out.items()}
    rec.update(deltas=np.diff(grid), grid=grid, knots=knots, way=way, mass=mass, grav=grav, cori=cori, taumax=taumax, fric=fric,
               vmax=vmax, amax=amax, sd_start=sd0, sd_end=sd1, scheme=np.array(1), kinds=np.array("vel,acc,torque"),
               differs=np.array(differs))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "nC", rec["a"].shape[2], "trajectories where the carried state changes bits:", differs, "of", B)


def dense_fixtures():
    dense_reuse_fixture()
    dense_fixture("dense_torque_d5_N40_collocation", 8, 5, 40, seed=71, kinds=("vel", "torque"), scheme=0)
    dense_fixture("dense_vel_acc_second_d6_N50", 6, 6, 50, seed=72, kinds=("vel", "acc", "second"), scheme=1, sd_mode="random")
    dense_fixture("dense_torque_only_d3_N30", 8, 3, 30, seed=73, kinds=("torque",), scheme=1, sd_mode="random")
    dense_fixture("dense_second_d7_N60_collocation", 4, 7, 60, seed=74, kinds=("vel", "acc", "second"), scheme=0)


def high_dof_fixtures():
    """round 3: the slim blocks of kernel family 3 (9..13 dof: 11 with boundary velocities, 13 with Collocation) and the
    two / three row slots per lane of family 4 above 16 dof (24, 32 dof)"""
    batch_fixture("batch_d11_N50_boundary", 8, 11, 50, seed=16, sd_mode="random", feasible=True)
    batch_fixture("batch_d13_N40_collocation", 6, 13, 40, seed=17, scheme=0, feasible=True)
    batch_fixture("batch_d24_N30", 4, 24, 30, seed=18)
    batch_fixture("batch_d32_N20_boundary", 3, 32, 20, seed=19, sd_mode="random")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--reuse-only" in sys.argv:
        reuse_fixture()
        pow_fixture()
        raise SystemExit(0)
    if "--reachable-only" in sys.argv:
        reachable_fixture()
        reachable_fixture("reach_d3_N40_collocation", B=16, d=3, N=40, seed=42, scheme=0)
        raise SystemExit(0)
    if "--param-only" in sys.argv:
        parametrizer_fixture()
        parametrizer_fixture("param_batch_d3_N400", B=6, d=3, N=400, seed=82)
        raise SystemExit(0)
    if "--dense-only" in sys.argv:
        dense_fixtures()
        raise SystemExit(0)
    if "--high-dof-only" in sys.argv:
        high_dof_fixtures()
        raise SystemExit(0)
    example_fixture()
    cpp_fixture()
    hard_fixture()
    sd_fixture()
    sd_fixture("sd_batch_d10_N50", B=12, d=10, N=50, seed=32)
    lp_fixture()
    batch_fixture("batch_d7_N200", 32, 7, 200, seed=20240924)
    batch_fixture("batch_d6_N500", 8, 6, 500, seed=20240925)
    batch_fixture("batch_d7_N100_boundary", 40, 7, 100, seed=77, sd_mode="random", feasible=True)
    batch_fixture("batch_d3_N60_collocation", 24, 3, 60, seed=5, scheme=0, feasible=True)
    batch_fixture("batch_d4_N80_acc_only", 16, 4, 80, seed=6, with_vel=False)
    batch_fixture("batch_d7_N120_tight", 24, 7, 120, seed=8, vscale=0.02, ascale=0.02, feasible=True)
    batch_fixture("batch_d5_N100_tiny_motion", 48, 5, 100, seed=9, wayscale=(-5.5, -0.5))
    # the dof range of the kernel families: 1, 2 (family 3 / 8 lanes), 9, 14, 16 (16 lanes per trajectory)
    batch_fixture("batch_d1_N40", 12, 1, 40, seed=11, feasible=True)
    batch_fixture("batch_d2_N50_boundary", 12, 2, 50, seed=12, sd_mode="random")
    batch_fixture("batch_d9_N60", 8, 9, 60, seed=13, feasible=True)
    batch_fixture("batch_d14_N40_boundary", 6, 14, 40, seed=14, sd_mode="random")
    batch_fixture("batch_d16_N30", 4, 16, 30, seed=15)
    high_dof_fixtures()
    parametrizer_fixture()
    parametrizer_fixture("param_batch_d3_N400", B=6, d=3, N=400, seed=82)
    dense_fixtures()
    reachable_fixture()
    reachable_fixture("reach_d3_N40_collocation", B=16, d=3, N=40, seed=42, scheme=0)
    reuse_fixture()
    pow_fixture()

"""Drive tests/host_cert/host_cert.cpp (the lane-level certificates of kernel family 3, compiled for the host, against the CPU
restatement of the reference) over workload families -- TEST INFRASTRUCTURE, CPU only.

    python tools/host_cert_hunt.py [family ...] [--rounds R] [--B n]

Families: natural, scaled, tight (velocity limits that bind), boundary (non-zero end velocities), collocation,
acc_only, sliver (three rows through one point to 1e-9 .. 1e-13: the family of tools/gpu_sliver_hunt.py), parallel (near-parallel
joints), feasible (compute_feasible_sets).  Prints one JSON line per workload; exit code 1 on any mismatch.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toppra_amd import batch  # noqa: E402  (host-side generator only)

EXE = os.environ.get("TPR_HOST_CERT_EXE") or os.path.join(tempfile.gettempdir(), "tpr_host_cert")
FLAG_VEL, FLAG_ACC, FLAG_INTERP = 1, 2, 4


def build(force=False):
    if os.environ.get("TPR_HOST_CERT_EXE"):   # a frozen copy (long hunts running beside a rebuild)
        return EXE
    srcs = [os.path.join(ROOT, "tests", "host_cert", "host_cert.cpp"), os.path.join(ROOT, "oracle", "seidel_oracle.c"),
            os.path.join(ROOT, "toppra_amd", "csrc", "tpr_cert_lane.hip.inc"), os.path.join(ROOT, "toppra_amd", "csrc", "tpr_device.hpp")]
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(s) for s in srcs):
        return EXE
    obj = EXE + "_oracle.o"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-c", "-o", obj, srcs[1]])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "tests", "host_cert", "hip_shim"),
                           "-o", EXE, srcs[0], obj, "-lm"])
    return EXE


def run(coef, breaks, grid, vlim, alim, sd_end=None, flags=FLAG_VEL | FLAG_ACC | FLAG_INTERP, mode=0, verbose=False, legacy=False, minform=False):
    build()
    B, _, nseg, d = coef.shape
    N = len(grid) - 1
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.array([B, d, nseg, N, flags, mode, 0 if sd_end is None else 1, (1 if legacy else 0) | (2 if minform else 0)], dtype=np.int32).tofile(f)
        for arr in (coef, breaks, grid, vlim, alim):
            np.ascontiguousarray(arr, dtype=np.float64).tofile(f)
        if sd_end is not None:
            np.ascontiguousarray(sd_end, dtype=np.float64).tofile(f)
        path = f.name
    try:
        pr = subprocess.run([EXE, path] + (["-v"] if verbose else []), capture_output=True, text=True)
    finally:
        os.remove(path)
    if os.environ.get("HOST_CERT_WHY"):
        print(pr.stderr.strip())
    lines = pr.stdout.strip().splitlines()
    for l in lines[:-1][:12]:
        print(l)
    return json.loads(lines[-1]), pr.returncode


def sliver_family(B, d, N, seed):
    """tools/gpu_sliver_hunt.py's generator with the base solution taken from the CPU restatement: acceleration limits of
    three joints moved so that their rows pass through the solution point of one stage to 1e-8 .. 1e-13."""
    from oracle import oracle as orc
    rng = np.random.default_rng(4200 + seed)
    data = batch.make_synthetic_batch(B, d, N, seed=4300 + seed)
    scale = 10.0 ** rng.uniform(-3, 1, size=(B, 1, 1, 1))
    scale[rng.random(B) < 0.5] = 1.0
    coef = data["coef"] * scale
    grid = data["grid"]
    base = orc.solve_batch(coef, data["breaks"], grid, data["vlim"], data["alim"], nthreads=0)
    ok = base["status"] == 0
    j = rng.integers(1, N - 1, size=B)
    rows = np.arange(B)
    u0 = np.where(ok, np.nan_to_num(base["u"][rows, j]), 0.0)
    x0 = np.where(ok, np.nan_to_num(base["sd2"][rows, j]), 0.5)
    off = np.where(rng.random(B) < 0.4, 0.0, 10.0 ** rng.uniform(-8, -2, size=B))
    u0 = u0 + off * rng.standard_normal(B) * np.maximum(1.0, np.abs(u0))
    x0 = np.maximum(x0 + off * rng.standard_normal(B) * np.maximum(1.0, np.abs(x0)), 0.0)
    qs = np.empty((B, d)); qss = np.empty((B, d))
    for b in range(B):
        q1, q2 = orc.path_eval(coef[b], data["breaks"], float(grid[j[b]]))
        qs[b], qss[b] = q1, q2
    alim = data["alim"].copy()
    joints = np.argsort(rng.random((B, d)), axis=1)[:, :3]
    for t in range(min(3, d)):
        k = joints[:, t]
        val = qs[rows, k] * u0 + qss[rows, k] * x0
        eps = 10.0 ** rng.uniform(-13, -8, size=B) * rng.choice([-1.0, 1.0], size=B) * np.maximum(1.0, np.abs(val))
        upper = rng.random(B) < 0.5
        width = 10 + 2 * rng.random(B)
        amax = np.where(upper, val + eps, val + eps + width)
        amin = np.where(upper, val + eps - width, val + eps)
        alim[rows, k, 0], alim[rows, k, 1] = amin, amax
    sd1 = np.where(rng.random(B) < 0.3, 0.2 * rng.random(B), 0.0)
    return coef, data["breaks"], grid, data["vlim"], alim, sd1


def workloads(family, B, seed):
    rng = np.random.default_rng(900 + seed)
    shapes = [(7, 200), (3, 60), (6, 120), (4, 80), (8, 64), (5, 100), (2, 40), (1, 50), (9, 50), (12, 40), (13, 30)]
    d, N = shapes[seed % len(shapes)]
    data = batch.make_synthetic_batch(B, d, N, seed=5000 + 17 * seed)
    coef, breaks, grid, vlim, alim = data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"]
    flags, mode, sd_end = FLAG_VEL | FLAG_ACC | FLAG_INTERP, 0, None
    if family == "scaled":
        coef = coef * 10.0 ** rng.uniform(-6, 0, size=(B, 1, 1, 1))
    elif family == "tight":
        vlim = vlim * 10.0 ** rng.uniform(-1.5, -0.3, size=(B, 1, 1))
    elif family == "boundary":
        sd_end = np.where(rng.random(B) < 0.7, rng.random(B) * 0.5, 0.0)
    elif family == "collocation":
        flags = FLAG_VEL | FLAG_ACC
    elif family == "acc_only":
        flags = FLAG_ACC | FLAG_INTERP
    elif family == "sliver":
        coef, breaks, grid, vlim, alim, sd_end = sliver_family(B, d, min(N, 64), seed)
    elif family == "parallel":
        # one joint a copy of another, scaled and tilted by 1e-14 .. 1e-6: two non-twin rows parallel to that accuracy
        if d >= 2:
            k0, k1 = 0, d - 1
            tilt = 10.0 ** rng.uniform(-14, -6, size=(B, 1, 1))
            sc = rng.uniform(0.5, 2.0, size=(B, 1, 1))
            coef = coef.copy()
            coef[:, :, :, k1] = sc * coef[:, :, :, k0] * (1.0 + tilt * rng.standard_normal((B, 4, coef.shape[2])))
            alim = alim.copy()
            alim[:, k1] = alim[:, k0] * sc[:, 0] * (1.0 + 10.0 ** rng.uniform(-12, -3, size=(B, 1)))
    elif family == "lower_ties":
        # two rows whose bounds on u at x = low1 (= 0 for these rest-to-rest problems: -c / a) agree to 1e-8 .. 1e-15 at one
        # stage: a near-tie between a prefix record of the lower-bound LP's cold run and the row visited next
        from oracle import oracle as orc
        alim = alim.copy()
        j = rng.integers(1, N - 1, size=B)
        for b in range(B):
            q1, q2 = orc.path_eval(coef[b], breaks, float(grid[j[b]]))
            k, m = rng.choice(d, size=2, replace=False) if d >= 2 else (0, 0)
            if d < 2 or q1[k] == 0 or q1[m] == 0:
                continue
            eps = 10.0 ** rng.uniform(-15, -8) * rng.choice([-1.0, 1.0])
            # bound of joint k: amax_k / |q1_k| (its + or - row, whichever has a > 0); make joint m's bound equal to it
            t_k = (alim[b, k, 1] if q1[k] > 0 else -alim[b, k, 0]) / abs(q1[k])
            val = t_k * abs(q1[m]) * (1.0 + eps)
            if q1[m] > 0:
                alim[b, m, 1] = val
            else:
                alim[b, m, 0] = -val
    elif family == "feasible":
        mode = 1
        if seed % 2:
            coef = coef * 10.0 ** rng.uniform(-4, 0, size=(B, 1, 1, 1))
    return (coef, breaks, grid, vlim, alim, sd_end, flags, mode), (d, N)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict((a[2:].split("=") + ["1"])[:2] for a in sys.argv[1:] if a.startswith("--"))
    rounds, B, start = int(opts.get("rounds", 1)), int(opts.get("B", 256)), int(opts.get("start", 0))
    legacy = "legacy" in opts
    minform = "minform" in opts  # the slides' verdicts as a running minimum (the TOPPRAsd kernels' form) instead of sign bits
    fams = args or ["natural", "scaled", "tight", "boundary", "collocation", "acc_only", "sliver", "parallel", "feasible"]
    build(force=start == 0)
    bad = 0
    for r in range(start, start + rounds):
        for fam in fams:
            for seed in range(11 * r, 11 * r + 11):
                (coef, breaks, grid, vlim, alim, sd_end, flags, mode), (d, N) = workloads(fam, B, seed)
                res, rc = run(coef, breaks, grid, vlim, alim, sd_end, flags, mode, legacy=legacy, minform=minform)
                res["family"] = fam
                res["seed"] = seed
                print(json.dumps(res), flush=True)
                bad += res["mismatch"]
    print("total mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Time the REFERENCE ITSELF (hungpham2511/toppra, Python + Cython seidel path) on the benchmark's synthetic batch, in the
build container (the GPU box has no /root/reference).  Writes profiles/r04_reference_cpu_rate.json, which bench.py quotes
in cpu_baseline.reference_itself (labelled as measured here, with date and command).

    python tools/time_reference.py [trajectories] [processes]
"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # one BLAS thread per worker process
    os.environ.setdefault(_v, "1")
import datetime
import json
import multiprocessing
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(args):
    lo, hi, d, N = args
    from oracle import ref_loader
    toppra = ref_loader.load()
    from toppra_amd import batch
    data = batch.make_synthetic_batch(hi, d, N)
    grid = data["grid"]
    t_total = t_solve = 0.0
    ok = 0
    for b in range(lo, hi):
        t0 = time.perf_counter()
        path = toppra.SplineInterpolator(data["knots"], data["waypoints"][b])
        cons = [toppra.constraint.JointVelocityConstraint(data["vlim"][b]),
                toppra.constraint.JointAccelerationConstraint(data["alim"][b])]
        inst = toppra.algorithm.TOPPRA(cons, path, gridpoints=grid, solver_wrapper="seidel")
        t1 = time.perf_counter()
        out = inst.compute_parameterization(0, 0)
        t2 = time.perf_counter()
        t_total += t2 - t0
        t_solve += t2 - t1
        ok += out[0] is not None
    return hi - lo, t_total, t_solve, ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    d, N = 7, 200
    # one core
    cnt, t_total, t_solve, ok = _worker((0, min(n, 64), d, N))
    single = {"trajectories": cnt, "end_to_end_traj_per_s": cnt / t_total, "compute_parameterization_only_traj_per_s": cnt / t_solve,
              "ok": ok}
    # multiprocessing.Pool over all cores
    chunks = [(i * n // procs, (i + 1) * n // procs, d, N) for i in range(procs)]
    t0 = time.perf_counter()
    with multiprocessing.Pool(procs) as pool:
        res = pool.map(_worker, chunks)
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    busy = sum(r[1] for r in res)
    out = {
        "what": "the reference itself: toppra.algorithm.TOPPRA(..., solver_wrapper='seidel').compute_parameterization(0, 0) on "
                "the benchmark's synthetic batch (7 dof, N = 200, velocity + acceleration (Interpolation))",
        "where": "build container (the GPU box has no /root/reference)", "date": datetime.date.today().isoformat(),
        "command": "python tools/time_reference.py %d %d" % (n, procs),
        "single_core": single,
        "pool": {"processes": procs, "trajectories": done, "wall_s": wall, "traj_per_s_wall_incl_pool_startup": done / wall,
                 "traj_per_s_sum_of_workers": procs * done / busy, "traj_per_s_per_core": done / busy},
        "waypoint_lps_per_s_per_core": 3 * N * done / busy,
    }
    path = os.path.join(ROOT, "profiles", "r04_reference_cpu_rate.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- batched TOPP-RA throughput on MI355X (driver contract: one JSON line on rank 0).

A "step" is one pass of the hot path (compute_constraint_params + backward controllable-set scan
+ forward parameterization scan, fused) over one batch of synthetic random 7-DoF splines that is
already resident in HBM.  Workload per GPU: B=65536 trajectories x N=200 gridpoints x 7 DoF,
velocity + acceleration (Interpolation) constraints, fp64 -- BASELINE.json's headline shape.
With --gpus N every rank solves its own 65536-trajectory shard (weak scaling; 8 ranks = config 5's
524288 trajectories) and the step ends with the RCCL gather of sd^2 to rank 0 that the north star
names.  `value` is whole-job trajectories/s.

  python bench.py                      # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(d, N, nseg):
    """Compulsory HBM traffic per trajectory of the fused path with full API outputs
    (SURVEY.md section 8(d)): inputs coef+breaks+limits+boundary velocities, outputs sd^2, u, K,
    status."""
    inp = 8 * (4 * nseg * d + (nseg + 1) + 4 * d + 2)
    out = 8 * ((N + 1) + N + 2 * (N + 1)) + 4
    return inp + out


def pmc_traffic(B, d, N, kernel_ms):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc.json, produced by tools/gpu_profile.sh + tools/pmc_summary.py --json).
    Only valid for the shape it was collected on; returns None otherwise."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))  # the latest round's passes
    if not cands:
        return None
    path = cands[-1]
    try:
        with open(path) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        return None
    if int(pmc["counters"].get("_Grid_Size", 0)) not in (B, B * 8) or (d, N) != (7, 200) or "hbm_bytes_per_launch" not in pmc:
        return None
    return {"bytes_per_launch": pmc["hbm_bytes_per_launch"],
            "bytes_per_launch_fetch_x2": pmc["hbm_bytes_per_launch_fetch_x2"],
            "gbps": pmc["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9,
            "kernel": pmc.get("kernel"), "valu_busy": pmc.get("valu_busy"), "avg_active_lanes": pmc.get("avg_active_lanes"),
            "valu_wave_instructions_per_launch": pmc["counters"].get("SQ_INSTS_VALU"),
            "salu_wave_instructions_per_launch": pmc["counters"].get("SQ_INSTS_SALU"),
            "REPLAYED": "every field of this block is read from profiles/%s -- rocprofv3 --pmc passes of this same command "
                        "(separate passes, no tracing), committed with the kernel they were taken on; they are NOT measured in "
                        "this run (counters cannot be read from inside bench.py).  Measured in this run: kernel_ms, ms_per_step, "
                        "value, roofline.achieved" % os.path.basename(path),
            "source": "profiles/%s (rocprofv3 --pmc, separate passes)" % os.path.basename(path)}


# fp64 VALU issue peak: 256 CUs x 4 SIMDs x 16 lanes per cycle at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
VALU_PEAK_LANE_INSTR = 256 * 4 * 16 * 2.4e9


PMC_PASSES = ("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE",
              "GRBM_GUI_ACTIVE FETCH_SIZE", "WRITE_SIZE GRBM_COUNT")


def pmc_measure(B, d, N, variant, kernel_ms, budget_s=240.0):
    """The same counters MEASURED IN THIS RUN (VERDICT r5 housekeeping): when rocprofv3 is on PATH, this bench re-runs itself
    under it -- short child runs of the headline step only, one per counter group (separate --pmc passes with --kernel-trace
    only, as the guide's HBM section prescribes) -- and reads the dominant kernel's averages from the counter CSVs.  Returns
    the block of pmc_traffic() with `measured_in_this_run`, or None (no rocprofv3, a failed pass, the time budget spent)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp or os.environ.get("TPR_BENCH_PMC_CHILD") or os.environ.get("TPR_BENCH_NO_PMC"):
        return None
    t0 = time.time()
    acc = {}
    kern = None
    env = dict(os.environ, TPR_BENCH_PMC_CHILD="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    with tempfile.TemporaryDirectory(prefix="tpr_pmc_") as tmp:
        for i, ctrs in enumerate(PMC_PASSES):
            if time.time() - t0 > budget_s:
                return None
            out = os.path.join(tmp, "p%d" % i)
            cmd = [rp, "--kernel-trace", "--pmc"] + ctrs.split() + ["-d", out, "-o", "run", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--kernel-reps", "1", "--batch", str(B), "--dof", str(d),
                   "--gridpoints", str(N), "--variant", str(variant), "--no-cpu-baseline", "--no-secondary", "--no-configs"]
            try:
                subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, timeout=max(30.0, budget_s - (time.time() - t0)), check=True)
            except (subprocess.SubprocessError, OSError):
                return None
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as fh:
                    for row in csv.DictReader(fh):
                        if "solve_kernel" not in row["Kernel_Name"] or int(float(row.get("Grid_Size", 0))) != B:
                            continue
                        kern = row["Kernel_Name"][:60]
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    if "FETCH_SIZE" not in avg or "WRITE_SIZE" not in avg:
        return None
    hbm = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024  # KiB per dispatch (MI355X_MICROARCH.md, HBM section)
    res = {"bytes_per_launch": hbm, "bytes_per_launch_fetch_x2": (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024,
           "gbps": hbm / (kernel_ms * 1e-3) / 1e9, "kernel": kern,
           "valu_wave_instructions_per_launch": avg.get("SQ_INSTS_VALU"), "salu_wave_instructions_per_launch": avg.get("SQ_INSTS_SALU"),
           "measured_in_this_run": "rocprofv3 --kernel-trace --pmc, %d separate passes of `bench.py --steps 3` as child processes "
                                   "(%.0f s); averages over the dominant kernel's dispatches" % (len(PMC_PASSES), time.time() - t0),
           "source": "measured in this run (rocprofv3 --pmc child passes)"}
    if "SQ_ACTIVE_INST_VALU" in avg and avg.get("GRBM_GUI_ACTIVE"):
        res["valu_busy"] = avg["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * avg["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_THREAD_CYCLES_VALU" in avg and avg.get("SQ_ACTIVE_INST_VALU"):
        res["avg_active_lanes"] = avg["SQ_THREAD_CYCLES_VALU"] / avg["SQ_ACTIVE_INST_VALU"]
    return res


def compute_roofline(pmc, kernel_ms):
    """What actually binds the fused kernel: vector-ALU instruction issue (SURVEY.md section 8(d) asks for it beside the
    HBM figure).  Lane-instructions per second = VALU wave-instructions per launch (PMC pass, replayed from
    profiles/) x 64 lanes / the kernel time measured in this run, against 16 lanes per SIMD and cycle."""
    if not pmc or not pmc.get("valu_wave_instructions_per_launch"):
        return None
    n = pmc["valu_wave_instructions_per_launch"]
    achieved = n * 64 / (kernel_ms * 1e-3)
    return {"bound": "valu_issue", "achieved": achieved, "peak": VALU_PEAK_LANE_INSTR, "unit": "fp64-lane-instructions/s",
            "frac": achieved / VALU_PEAK_LANE_INSTR,
            "valu_wave_instructions_per_launch": n, "kernel_ms": kernel_ms,
            "valu_busy_pmc": pmc.get("valu_busy"), "avg_active_lanes_pmc": pmc.get("avg_active_lanes"),
            "instruction_count_source": pmc["source"] + (" -- kernel_ms is this run's too" if pmc.get("measured_in_this_run") else " -- REPLAYED, not measured in this run; kernel_ms is this run's"),
            "note": "every lane-instruction of this kernel is an fp64 / integer VALU operation of a 64-wide wave (no MFMA: the "
                    "path has no dense contraction); one wave per SIMD, so the issue rate is additionally capped by the "
                    "single-wave issue interval (~5 cycles per instruction of any kind: profiles/r02_single_wave_issue_microbench.log, round 2)"}


def cpu_baseline(data, target_seconds=12.0):
    """The oracle (C port of the reference's seidel path, bit-exact with it) on the host cores, on a bounded
    sample of the same workload: one thread, then one thread per physical core (bound), with the parallel
    efficiency of the two.  Every worker thread reuses one arena for all of its trajectories
    (oracle/seidel_oracle.c: a fresh wrapper per trajectory used to mean fresh zero pages from the C library,
    which is what a 256-thread run then measured -- 7 % efficiency in round 2)."""
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        physical = os.cpu_count() or 1
    logical = os.cpu_count() or physical
    # What this process may actually use: its affinity mask and the CPU bandwidth of its cgroup (the GPU box hands the
    # container 16 CPUs' worth of a 128-core host: more threads than that are throttled and scale NEGATIVELY --
    # tools/cpu_scaling_probe.py: 16 threads 15.3x one thread, 128 threads 9.3x)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q = float(fh.read())
                quota = None if q <= 0 else q / float(fp.read())
        except (OSError, ValueError):
            quota = None
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    cores = max(1, min(physical, affinity, int(quota) if quota else physical))
    os.environ.setdefault("OMP_PLACES", "cores")     # (read when libgomp is loaded, i.e. by the import below)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    from oracle import oracle as orc
    B = data["coef"].shape[0]

    def run(n, threads):
        t0 = time.perf_counter()
        orc.solve_batch(data["coef"][:n], data["breaks"], data["grid"], data["vlim"][:n], data["alim"][:n], nthreads=threads)
        return time.perf_counter() - t0

    # one thread: ~2 s
    run(64, 1)
    t = run(256, 1)
    n1 = int(min(B, max(256, 256 * 2.0 / max(t, 1e-6))))
    t1 = min(run(n1, 1), run(n1, 1))
    single = n1 / t1
    # one thread per physical core
    run(min(B, 8 * cores), cores)
    n0 = min(B, 64 * cores)
    t = run(n0, cores)
    n = int(min(B, max(n0, n0 * 2.0 / max(t, 1e-6))))  # ~2 s per pass, repeated to the target
    reps, total, best = 0, 0.0, 1e30
    while total < target_seconds and reps < 64:
        dt = run(n, cores)
        total += dt
        best = min(best, dt)
        reps += 1
    allcore = n * reps / total
    # BASELINE config 1 beside the GPU's latency figure: one trajectory, one thread
    c1 = {}
    for label, N in (("N100", 100), ("N289", 289)):
        from toppra_amd import batch as tb
        one = tb.make_synthetic_batch(1, 7, N, seed=9)
        args = (one["coef"], one["breaks"], one["grid"], one["vlim"], one["alim"])
        orc.solve_batch(*args, nthreads=1)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            orc.solve_batch(*args, nthreads=1)
            ts.append(time.perf_counter() - t0)
        c1[label + "_ms"] = float(np.median(ts) * 1e3)
    port = {"value": allcore, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "single_thread": {"value": single, "unit": "trajectories/s", "sample": "%d trajectories" % n1},
            "parallel_efficiency": allcore / (cores * single),
            "best_pass": n / best,
            "host": {"physical_cores": physical, "logical_cpus": logical, "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
                     "note": "`cores` = threads used = min(physical cores, affinity, cgroup CPU quota): what the container may run at once"},
            "thread_binding": "OMP_PLACES=%s OMP_PROC_BIND=%s" % (os.environ.get("OMP_PLACES"), os.environ.get("OMP_PROC_BIND")),
            "config1_single_trajectory": dict(c1, note="oracle/seidel_oracle.c on ONE host thread, one 7-dof trajectory per call "
                                              "(N = 100 / 289 gridpoints): what configs.C1_single_trajectory is up against"),
            "sample": "first %d trajectories of the rank-0 batch x %d passes (%.1f s), oracle/seidel_oracle.c "
                      "(C restatement of the reference's seidel path, bit-exact with it) with OpenMP on %d threads (bound, one "
                      "per core: every CPU this container is allowed); per-thread arenas" % (n, reps, total, cores)}
    # The headline baseline is the reference's own compiled solver when its binary travelled with the snapshot
    # (oracle/_ref: kind "reference"), with the C port -- ~18x faster per core: no Python call per stage LP -- beside it;
    # without the binary the port is the baseline, as in rounds 1-3.
    ref = reference_solver_rate(data, cores, orc)
    # Top level (value / unit / cores / kind / sample): the reference's own compiled solver when its binary travelled with the
    # snapshot, else the port.  BOTH are always reported under fixed keys -- cpu_baseline.port and
    # cpu_baseline.reference_solver -- so that rounds and machines stay comparable whatever the top level holds.
    out = dict(ref) if "value" in ref else dict(port)
    out["port"] = port
    out["reference_solver"] = ref
    if "value" in ref:
        out["port_over_reference_solver"] = port["value"] / ref["value"]
    out["per_config"] = config_baselines(cores, orc, port, ref)
    out["reference_itself"] = reference_rate()
    return out


def config_baselines(cores, orc, port, ref):
    """CPU rates beside BASELINE configs 2 and 3 ("the oracle rate for configs 1-3 and the headline shape";
    config 1 is cpu_baseline.port.config1_single_trajectory): the same two solvers, the same harness, on the configs' own
    shapes.  Config 2 (4096 x 7 x 200) has the headline's shape per trajectory: its rates are the headline's; config 3
    (6 dof, N = 500) is timed here on a bounded sample."""
    from toppra_amd import batch as tb
    out = {"C2_batch4096_d7_N200": {
        "port": {"value": port["value"], "unit": "trajectories/s", "cores": cores, "seconds_for_the_batch": 4096 / port["value"]},
        "reference_solver": ({"value": ref["value"], "unit": "trajectories/s", "cores": cores, "seconds_for_the_batch": 4096 / ref["value"]}
                             if "value" in ref else ref),
        "note": "same shape per trajectory as the headline batch: the headline's rates (same run)"}}
    try:
        n = 24 * cores
        data3 = tb.make_synthetic_batch(n, 6, 500)
        orc.solve_batch(data3["coef"][:cores], data3["breaks"], data3["grid"], data3["vlim"][:cores], data3["alim"][:cores], nthreads=cores)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            orc.solve_batch(data3["coef"], data3["breaks"], data3["grid"], data3["vlim"], data3["alim"], nthreads=cores)
            reps += 1
        rate_port = n * reps / (time.perf_counter() - t0)
        c3 = {"port": {"value": rate_port, "unit": "trajectories/s", "cores": cores, "seconds_for_the_batch": 65536 / rate_port,
                       "sample": "%d trajectories x %d passes, oracle/seidel_oracle.c on %d threads" % (n, reps, cores)}}
        if "value" in ref:
            from oracle import ref_solver_baseline as rb
            r3 = rb.time_passes(data3, 24 * cores, cores)  # (its untimed setup, Python per gridpoint, is what bounds the sample)
            c3["reference_solver"] = {"value": r3["trajectories_per_s"], "unit": "trajectories/s", "cores": cores,
                                      "seconds_for_the_batch": 65536 / r3["trajectories_per_s"],
                                      "sample": "%d trajectories on %d processes (%.2f s of solver passes): the reference's compiled "
                                                "seidelWrapper under the restated passes" % (r3["trajectories"], cores, r3["seconds"])}
        else:
            c3["reference_solver"] = ref
        out["C3_batch65536_d6_N500"] = c3
    except Exception as exc:  # noqa: BLE001
        out["C3_batch65536_d6_N500"] = {"error": repr(exc)[:200]}
    return out


def reference_solver_rate(data, cores, orc):
    """kind "reference", timed in THIS run: the reference's own compiled seidel solver (oracle/_ref, built from the sources
    where they lie under /root/reference and shipped to the GPU box as a binary) driven by the reference's two passes
    restated as Python loops (oracle/ref_solver_baseline.py) -- 3 N Python -> Cython crossings per trajectory, as in the
    reference.  One process, then one process per usable core; the solver passes only (constraint parameters and the
    wrapper's row build are outside the timed region).  Its results on the sample are compared with the port's."""
    try:
        from oracle import ref_solver_baseline as rb
        if not rb.available():
            return {"error": "oracle/_ref holds no compiled reference solver (it is built where /root/reference exists)"}
        n1, per = 384, 256
        one = rb.time_passes(data, n1, 1)
        allp = rb.time_passes(data, min(data["coef"].shape[0], per * cores), cores)
        # the same bits as the port (and therefore as the GPU) on a few trajectories
        same = True
        want = orc.solve_batch(data["coef"][:4], data["breaks"], data["grid"], data["vlim"][:4], data["alim"][:4], nthreads=1)
        for k in range(4):
            vel, acc = rb.constraint_tuples(data["coef"][k], data["breaks"], data["grid"], data["vlim"][k], data["alim"][k])
            w = rb.make_wrapper([rb.PrecomputedConstraint(vel, False), rb.PrecomputedConstraint(acc, True)], None, data["grid"])
            sdd, sd, K = rb.parameterization(w, 0.0, 0.0)
            same &= sd is not None and bool(np.array_equal(sd, np.sqrt(want["sd2"][k])) and np.array_equal(sdd, want["u"][k]) and np.array_equal(K, want["K"][k]))
        return {"value": allp["trajectories_per_s"], "unit": "trajectories/s", "cores": cores, "kind": "reference",
                "single_process": {"value": one["trajectories_per_s"], "sample": "%d trajectories" % n1},
                "parallel_efficiency": allp["trajectories_per_s"] / (cores * one["trajectories_per_s"]),
                "identical_bits_to_the_port_on_4_trajectories": same,
                "sample": "first %d trajectories of the rank-0 batch on %d processes (%.2f s of solver passes in the slowest one): the "
                          "reference's compiled cy_seidel_solverwrapper.seidelWrapper (oracle/_ref), solve_lp1d=True, under the two passes "
                          "of TOPPRA.compute_parameterization restated in Python (reachability_algorithm.py:166-238, 240-376); "
                          "compute_constraint_params and the wrapper's construction are not timed" % (allp["trajectories"], cores, allp["seconds"])}
    except Exception as exc:  # noqa: BLE001  (a measurement leg must not take the bench line down)
        return {"error": repr(exc)}


def reference_rate():
    """The reference's OWN rate (Python + Cython seidel path), measured in the build container by tools/time_reference.py
    and committed under profiles/ -- REPLAYED here: the GPU box has no /root/reference."""
    import glob
    note = ("the reference cannot run on the GPU box: /root/reference is absent there and its sources may not be copied into "
            "this repository, so the baseline timed in THIS run is the C port (~70x faster per core than the reference: no "
            "Python call overhead -- a stronger baseline); the figures of this block were measured in the build container")
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_rate.json")))
    if not cands:
        return {"value": 180.0, "unit": "trajectories/s per core", "where": "build container, 1 core (round 1 probe)", "note": note}
    with open(cands[-1]) as fh:
        ref = json.load(fh)
    return {"value": ref["pool"]["traj_per_s_per_core"], "unit": "trajectories/s per core (end to end: spline, constraints, TOPPRA, "
            "compute_parameterization)", "all_cores": {"value": ref["pool"]["traj_per_s_sum_of_workers"], "processes": ref["pool"]["processes"]},
            "compute_parameterization_only_single_core": ref["single_core"]["compute_parameterization_only_traj_per_s"],
            "where": ref["where"], "date": ref["date"], "command": ref["command"],
            "REPLAYED": "from profiles/%s, not measured in this run" % os.path.basename(cands[-1]), "note": note}


TOL_PROBE = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from toppra_amd import batch as tb
data = tb.make_synthetic_batch(%(B)d, %(d)d, %(N)d, seed=%(seed)d)
dev = torch.device("cuda", 0)
dv = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]
out = tb.solve_batch(*dv); torch.cuda.synchronize()
ms = tb.solve_batch_timed(*dv, out, reps=5)
np.save(%(tmp)r, out["sd2"].cpu().numpy()); np.save(%(tmp)r + ".status.npy", out["status"].cpu().numpy())
print(json.dumps({"kernel_ms": ms}))
"""


NOTE_SOUND_TOLERANCE = (
    "NOT the product: the second measurement build (python -m toppra_amd.build --sound-tolerance, -DTPR_SOUND_TOLERANCE; VERDICT r4 item 3) "
    "-- the product's SOUND, trace-following certificates, returning the verified vertex from a reciprocal estimate instead of "
    "replicating the reference's last pivot (no cross-product guard, no IEEE divisions there) and reciprocal quotients in the forward "
    "1-variable LP; the cooperative batches run the reference's iteration exactly as in the product.  product - this = what "
    "BIT-EXACTNESS costs once soundness is kept; this - tolerance_build = what SOUNDNESS costs.  "
    "tools/gpu_sound_tolerance_report.py (profiles/r05_sound_tolerance_report.json): statuses and NaN patterns identical to the "
    "product on the headline, irregular, scaled, sliver and near-parallel families, max |d sd^2| 2.1e-11.  Not exposed as a flag: the bar "
    "set for that (<= 1.6 ms) is not met")


def tolerance_probe(B, d, N, seed, sd2, status, libname="libtoppra_hip_tol.so", note=None):
    """The opt-in measurement build (toppra_amd/libtoppra_hip_tol.so, python -m toppra_amd.build --tolerance): same
    sources with the certified vertices returned as they are, contracted multiply-adds and reciprocal division.
    Run in a subprocess (one library per process) on the headline batch; deviation against the product's result."""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "toppra_amd", libname)
    if not os.path.exists(lib):
        return None
    tmp = os.path.join(tempfile.gettempdir(), "tpr_tol_probe_%d.npy" % os.getpid())
    env = dict(os.environ, TOPPRA_HIP_LIB=lib)
    try:
        lines, outp = [], None
        for _attempt in range(2):  # (one retry: the probe is a second process on a GPU this one still holds memory on)
            outp = subprocess.run([sys.executable, "-c", TOL_PROBE % dict(root=ROOT, B=B, d=d, N=N, seed=seed, tmp=tmp)], env=env,
                                  capture_output=True, text=True, timeout=300)
            lines = [l for l in outp.stdout.splitlines() if l.startswith("{")]
            if lines:
                break
        if not lines:
            return {"error": "the probe process printed no result (exit code %s): %s" % (outp.returncode, outp.stderr.strip()[-400:])}
        ms = json.loads(lines[-1])["kernel_ms"]
        tol_sd2, tol_status = np.load(tmp), np.load(tmp + ".status.npy")
    except Exception as exc:  # a failed probe must not take the bench line with it
        return {"error": repr(exc)[:200]}
    finally:
        for f in (tmp, tmp + ".status.npy"):
            if os.path.exists(f):
                os.remove(f)
    return {"kernel_ms": ms, "value_per_gpu": B / ms * 1e3, "unit": "trajectories/s",
            "max_abs_dsd2_vs_product": float(np.nanmax(np.abs(tol_sd2 - sd2))),
            "status_identical": bool(np.array_equal(tol_status, status)),
            "nan_pattern_identical": bool(np.array_equal(np.isnan(tol_sd2), np.isnan(sd2))),
            "note": note or "NOT the product: measurement build answering 'what do the reference's bits cost' (the product predicts the "
                    "reference's whole pivot trace before it answers an LP from a certificate and replicates its last-pivot "
                    "arithmetic FMA-free with correctly rounded divisions; this build certifies the final vertex only -- round 3's "
                    "certificates, which return an optimum where a sliver pivot ends the reference's run -- and returns that vertex "
                    "itself, with contracted multiply-adds and reciprocal division).  The north star's bar is "
                    "1e-8 on sd^2; tools/gpu_tolerance_report.py checks every fixture (profiles/r05_tolerance_report.json)"}


def baseline_configs(torch, tb, dev):
    """Every configuration of BASELINE.json.configs in one place (rank 0, N=1, outside the timed region):
    C1 latency through the drop-in class, C2 / C3 / C4 kernel times with inputs resident in HBM, and the
    PCIe-inclusive time of the host-buffer entry for the headline shape."""
    import toppra_amd as ta

    def dev_args(data):
        return [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("coef", "breaks", "grid", "vlim", "alim")]

    def kernel(B, d, N, reps=5):
        data = tb.make_synthetic_batch(B, d, N)
        dv = dev_args(data)
        out = tb.solve_batch(*dv)
        torch.cuda.synchronize()
        ms = tb.solve_batch_timed(*dv, out, reps=reps)
        nseg = data["coef"].shape[2]
        gbs = algorithmic_bytes(d, N, nseg) * B / (ms * 1e-3) / 1e9
        return {"batch": B, "dof": d, "gridpoints": N, "kernel_ms": ms, "trajectories_per_s": B / ms * 1e3,
                "waypoint_lps_per_s": 3 * N * B / ms * 1e3, "ok_fraction": float((out["status"] == 0).double().mean().item()),
                "algorithmic_bytes_per_trajectory": algorithmic_bytes(d, N, nseg), "achieved_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}

    def wall(fn, reps):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts) * 1e3)

    res = {}
    # C1: examples/plot_kinematics.py -- one 7-dof spline through the reference's own class surface
    np.random.seed(9)
    way = np.random.randn(5, 7)
    path = ta.SplineInterpolator(np.linspace(0, 1, 5), way)
    vlim_ = 10 + np.random.rand(7) * 20
    alim_ = 10 + np.random.rand(7) * 2
    cons = [ta.constraint.JointVelocityConstraint(np.vstack((-vlim_, vlim_)).T),
            ta.constraint.JointAccelerationConstraint(np.vstack((-alim_, alim_)).T)]
    c1 = {}
    for label, grid in (("auto_grid", None), ("N100", np.linspace(0, 1, 101))):
        inst = ta.algorithm.TOPPRA(cons, path, gridpoints=grid)
        ms = wall(lambda: inst.compute_parameterization(0, 0), 20)
        c1[label] = {"gridpoints": int(len(inst.problem_data.gridpoints) - 1), "compute_parameterization_ms": ms,
                     "return_code": str(inst.problem_data.return_code.name)}
    c1["note"] = ("single trajectory, host arrays in and out through toppra_amd.algorithm.TOPPRA (one launch per pass); "
                  "latency floor of a 1-trajectory launch, not a throughput figure.  Reference (Python+Cython seidel, "
                  "build container, 1 core): ~3.9 ms for the auto grid")
    res["C1_single_trajectory"] = c1
    res["C2_batch4096_d7_N200"] = kernel(4096, 7, 200)
    res["C3_batch65536_d6_N500"] = kernel(65536, 6, 500, reps=3)
    data4 = tb.make_synthetic_batch(16384, 7, 100)
    dv4 = dev_args(data4)
    ell = [1e-3, 5e-2, 9e-3]  # examples/plot_robust_kinematics.py:26-28
    ms4 = wall(lambda: tb.robust_solve_batch(*dv4, ell), 3)
    res["C4_robust_batch16384_d7_N100"] = {
        "batch": 16384, "dof": 7, "gridpoints": 100, "ms": ms4, "trajectories_per_s": 16384 / ms4 * 1e3,
        "note": "RobustLinearConstraint over JointAcceleration, ellipsoid (1e-3, 5e-2, 9e-3); the reference's ECOS stage "
                "problems solved exactly; parity unpinned against ECOS (absent), cross-checked at 1e-7 against an independent exact solver (tests/test_gpu_robust.py)"}
    res["C5_batch524288_8gpu"] = "this bench with --gpus 8 (65536 trajectories per rank + RCCL gather of sd^2)"
    res["large_batch262144_d7_N200"] = dict(kernel(262144, 7, 200, reps=3),
                                            note="four rounds of one wave per SIMD (a two-waves-per-block form was built and measured in round 4: slower, DESIGN.md 4.1)")
    res["d12_batch65536_N200"] = dict(kernel(65536, 12, 200, reps=3), note="12 dof: family 3 with slim blocks and the trace-following certificates (DESIGN.md 3.2); rows across lanes: 12.1 ms")
    # dense rows (any canonical-linear constraint list): the headline problem's own rows materialised as
    # seidelWrapper would hold them (144 KB per trajectory) and solved from those arrays -- the HBM-heavy form of the path
    datad = tb.make_synthetic_batch(65536, 7, 200)
    dvd = dev_args(datad)
    rows = tb.constraint_params_batch(*dvd)
    dense = (rows["a"], rows["b"], rows["c"], rows["low"], rows["high"], dvd[2][1:] - dvd[2][:-1])
    same = bool(torch.equal(torch.nan_to_num(tb.solve_dense_batch(*dense)["sd2"], nan=-7.0),
                            torch.nan_to_num(tb.solve_batch(*dvd)["sd2"], nan=-7.0)))
    msd = wall(lambda: tb.solve_dense_batch(*dense), 3)
    row_bytes = 2 * 8 * 65536 * (201 * (3 * 30 + 4) + 200)  # rows + boxes + deltas, read by the backward and by the forward scan
    res["dense_rows_batch65536_d7_N200"] = {
        "batch": 65536, "rows_per_stage": 30, "gridpoints": 200, "ms": msd, "trajectories_per_s": 65536 / msd * 1e3,
        "row_bytes_read": row_bytes, "GBps": row_bytes / (msd * 1e-3) / 1e9, "hbm_frac_of_8TBps_peak": row_bytes / (msd * 1e-3) / 8e12,
        "identical_bits_to_the_fused_path": same,
        "note": "tpr_solve_dense_batch: every stage LP through the reference's full Seidel iteration on rows read from HBM "
                "(the entry that serves torque / second-order / hand-written constraints); bound by the iteration's instruction "
                "count like the fused strict mode, the row traffic hides behind it"}
    del rows, dense, dvd
    # PCIe-inclusive: numpy in -> numpy out through the host-buffer entry (H2D 59 MB, kernel, D2H)
    datah = tb.make_synthetic_batch(65536, 7, 200)
    hargs = [datah[k] for k in ("coef", "breaks", "grid", "vlim", "alim")]
    res["host_buffers_65536x7x200"] = {
        "all_outputs_ms": wall(lambda: tb.solve_batch(*hargs), 3),
        "without_K_ms": wall(lambda: tb.solve_batch(*hargs, want_K=False), 3),
        "sd2_only_ms": wall(lambda: tb.solve_batch(*hargs, want_K=False, want_u=False), 3),
        "note": "PCIe-inclusive wall time of tpr_solve_batch with host pointers (page-locked result arrays); never `value`.  sd2_only: neither K nor u returned -- all that retiming (compute_trajectory) reads"}
    return res


def end_to_end(torch, tb, dev, B=65536, d=7, N=200, samples=64):
    """Waypoints -> q(t) with everything on the device, the headline shape (rank 0, N=1, outside the timed region):
    tpr_spline_fit_batch (SplineInterpolator.__init__) -> tpr_solve_batch (compute_parameterization; sd only: what
    retiming reads) -> tpr_param_spline_batch (ParametrizeSpline, the reference's default output parametrizer) ->
    tpr_ppoly_eval_batch (q at `samples` times per trajectory).  Milliseconds per stage with HIP events on torch's
    current stream (the library launches on it), median of 5; algorithmic bytes per stage beside them."""
    rng = np.random.default_rng(20240924)
    way = torch.from_numpy(rng.standard_normal((B, 5, d))).to(dev)
    knots = torch.linspace(0, 1, 5, dtype=torch.float64, device=dev)
    grid = torch.linspace(0, 1, N + 1, dtype=torch.float64, device=dev)
    vlim_ = 10 + 20 * rng.random((B, d))
    alim_ = 10 + 2 * rng.random((B, d))
    vlim = torch.from_numpy(np.stack((-vlim_, vlim_), axis=-1)).to(dev)
    alim = torch.from_numpy(np.stack((-alim_, alim_), axis=-1)).to(dev)
    state = {}

    def fit():
        state["coef"], state["breaks"] = tb.spline_fit_batch(knots, way)

    def solve():
        state["sol"] = tb.solve_batch(state["coef"], state["breaks"], grid, vlim, alim, want_sd=True, want_K=False, want_u=False)

    def param():
        state["sp"] = tb.param_spline_batch(state["coef"], state["breaks"], grid, state["sol"]["sd"])

    def evaluate():
        sp = state["sp"]
        dur = sp["knot_times"].gather(1, (sp["counts"].long() - 1).clamp(min=0).unsqueeze(1))
        state["times"] = torch.linspace(0, 1, samples, dtype=torch.float64, device=dev).unsqueeze(0) * dur
        state["q"] = tb.ppoly_eval_batch(sp["coef"], sp["knot_times"], state["times"], 0, sp["counts"])

    def sample_fused():
        state["qs"] = tb.param_spline_sample_batch(state["coef"], state["breaks"], grid, state["sol"]["sd"], frac, orders=(0,))

    frac = torch.linspace(0, 1, samples, dtype=torch.float64, device=dev)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    nseg = 4
    stages = {}
    for name, fn, nbytes in (
            ("spline_fit", fit, 8 * (B * 5 * d + B * 4 * nseg * d)),
            ("solve_sd_only", solve, B * (8 * (4 * nseg * d + 4 * d) + 8 * 2 * (N + 1) + 4)),
            ("param_spline", param, 8 * B * ((N + 1) + 4 * nseg * d + 4 * N * d + (N + 1)) + 4 * B),
            ("ppoly_eval_%d_samples" % samples, evaluate, 8 * B * samples * (1 + d + 4 * d) + 8 * B * (N + 1))):
        ms = timed(fn)
        gbps = nbytes / (ms * 1e-3) / 1e9
        stages[name] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": gbps, "hbm_frac_of_8TBps_peak": gbps / 8000.0,
                        "hbm_frac_of_6.3TBps_achievable": gbps / 6300.0}
    total = sum(s["ms"] for s in stages.values())
    # the same samples without the coefficient table in between (tpr_param_spline_sample_batch): fit + evaluation in one
    # launch from LDS-resident knot derivatives
    ms_fused = timed(sample_fused)
    nb_fused = 8 * B * ((N + 1) + 4 * nseg * d + samples * d) + 8 * samples
    same = bool(torch.equal(torch.nan_to_num(state["qs"]["q"], nan=-7.0), torch.nan_to_num(state["q"], nan=-7.0)))
    fused = {"ms": ms_fused, "algorithmic_bytes": nb_fused, "GBps": nb_fused / (ms_fused * 1e-3) / 1e9,
             "identical_bits_to_table_plus_evaluation": same,
             "total_ms_with_it": stages["spline_fit"]["ms"] + stages["solve_sd_only"]["ms"] + ms_fused}
    ok = float((state["sol"]["status"] == 0).double().mean().item())
    finite = bool(torch.isfinite(state["q"][state["sol"]["status"] == 0]).all().item())
    return {"workload": "batch=%d, %d-DoF, 5 waypoints -> N=%d gridpoints -> %d samples of q(t) per trajectory; device-resident" % (B, d, N, samples),
            "stages": stages, "total_ms": total, "trajectories_per_s": B / total * 1e3, "ok_fraction": ok, "q_finite_where_ok": finite,
            "sampled_without_the_table": fused, "trajectories_per_s_without_the_table": B / fused["total_ms_with_it"] * 1e3,
            "note": "param_spline writes the [B, 4, N, d] coefficient table (%.2f GB): the one HBM-bound stage of the pipeline "
                    "(6.3 TB/s achievable: MI355X_MICROARCH.md); ppoly_eval's bytes are what its samples need (a time, four coefficient "
                    "rows of d doubles and d outputs per sample, the breakpoints once): %d samples read a fraction of the table, at "
                    "cache-line granularity" % (8 * B * 4 * N * d / 1e9, samples)}


def self_launch(n):
    """Re-run this command line under torch.distributed.run with n ranks on this node; returns its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU")
    ap.add_argument("--dof", type=int, default=7)
    ap.add_argument("--gridpoints", type=int, default=200, help="N (stages)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=5)
    ap.add_argument("--gather", choices=["auto", "overlap", "sequential"], default="auto",
                    help="N > 1: where the gather of sd^2 goes relative to the next solve (auto: timed during warm-up)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="multi-rank control-flow rehearsal on ONE GPU: every rank uses cuda:0 and the gather runs over "
                         "gloo on host copies (RCCL refuses two ranks on one device); timings are meaningless")
    ap.add_argument("--stub-solver", action="store_true",
                    help="control-flow rehearsal WITHOUT a GPU (tests/test_distributed_gloo.py, world size 8 over gloo): "
                         "the solve is replaced by a stub that tags sd^2 with (rank, step) so that rank 0 can check what "
                         "the gather delivered; everything else -- gather placement calibration, pipelined gather, "
                         "max-over-ranks timing, per-rank kernel times, the JSON line -- is the code of a real run.  "
                         "Timings are meaningless")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs block (C1-C4, host-buffer path)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary (full iteration, family 2) measurements (used under rocprofv3 so that the "
                         "kernel statistics cover the headline mode only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves, exactly as the driver's own command would
        # (one process per GPU over RCCL, 127.0.0.1 rendezvous on a free port); rank 0 of the child job prints the line
        raise SystemExit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    stub = args.stub_solver
    if stub:
        args.rehearsal = True
        args.no_secondary = args.no_configs = args.no_cpu_baseline = True
    else:
        from toppra_amd import build as hip_build
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            hip_build.ensure_built()  # no-op when the in-tree library travelled with the snapshot
        else:
            for _ in range(600):      # the other ranks wait for rank 0's build instead of racing it
                if os.path.exists(hip_build.LIB):
                    break
                time.sleep(0.5)
    from toppra_amd import batch as tb

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and "WORLD_SIZE" in os.environ:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if args.rehearsal:
        local_rank = 0
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None  # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if (args.rehearsal and world > 1) else dev  # where collectives run

    B, d, N = args.batch, args.dof, args.gridpoints
    if stub:
        nseg, dv, data = 4, None, None
    else:
        data = tb.make_synthetic_batch(B, d, N, seed=20240924 + rank)
        nseg = data["coef"].shape[2]
        dv = {k: torch.from_numpy(np.ascontiguousarray(data[k])).to(dev)
              for k in ("coef", "breaks", "grid", "vlim", "alim")}
    stub_step = {"n": 0}

    def solve_once():
        if stub:  # sd^2 tagged with (rank, step): what rank 0 must find in its receive buffers after the gather
            stub_step["n"] += 1
            tag = float(1000 * rank + stub_step["n"])
            return {"sd2": torch.full((B, N + 1), tag, dtype=torch.float64), "status": torch.zeros(B, dtype=torch.int32), "tag": tag}
        return tb.solve_batch(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], variant=args.variant)

    # N > 1: the only communication is the gather of sd^2 to rank 0 (RCCL); it is issued asynchronously so
    # that step k's gather rides the xGMI links while step k+1 is being solved (toppra_amd/distributed.py)
    from toppra_amd.distributed import PipelinedGather
    gatherer = PipelinedGather(B, N + 1, torch.float64, cdev) if world > 1 else None

    # Two ways to place the gather: "overlap" (step k's gather rides the links while step k+1 is being solved)
    # or "sequential" (the next solve is ordered after the gather).  The solve kernel is sized for exactly one
    # wave per SIMD, so if the collective's own kernels occupy compute units while it runs, the displaced
    # blocks need a second round and overlapping LOSES; which one wins depends on RCCL's channel count on the
    # node.  "auto" times both during the warm-up steps (all ranks agree through a MAX all-reduce) and keeps
    # the faster one for the timed region.
    gather_mode = {"value": "overlap" if args.gather == "auto" else args.gather}

    def step():
        out = solve_once()
        if gatherer is not None:
            gatherer.submit(out["sd2"].to(cdev))
            if gather_mode["value"] == "sequential":
                gatherer.order_after()  # stream-level: the next solve waits for this gather (no host block)
        return out

    def fence():
        if gatherer is not None:
            gatherer.finish()  # the last step's gather is inside the timed region
        sync()
        if world > 1:
            dist.barrier()
        sync()

    out = None
    calibration = None
    if world > 1 and args.gather == "auto":
        # (at least two steps per placement -- one untimed, one timed -- even when fewer warm-up steps were asked
        # for: extra untimed steps do not touch the timed region)
        timings = {}
        for mode, n in (("overlap", max(2, (args.warmup + 1) // 2)), ("sequential", max(2, args.warmup // 2))):
            gather_mode["value"] = mode
            out = step()  # first step of a mode is not timed (buffers, lazy init)
            fence()
            tc = time.perf_counter()
            for _ in range(max(1, n - 1)):
                out = step()
            fence()
            tm = torch.tensor([(time.perf_counter() - tc) / max(1, n - 1)], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            timings[mode] = float(tm.item()) * 1e3
        gather_mode["value"] = min(timings, key=timings.get)
        calibration = {"ms_per_step": timings, "chosen": gather_mode["value"]}
    else:
        for _ in range(args.warmup):
            out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_check = None
    if stub and gatherer is not None and rank == 0:
        # after the last fence the receive buffers hold every rank's LAST step: tag = 1000 r + (number of steps it ran)
        bufs = gatherer.bufs
        gather_check = all(bool((bufs[r] == float(1000 * r + stub_step["n"])).all()) for r in range(world))
    # dominant kernel: average launch duration with HIP events on the launch stream
    if stub:
        kernel_ms = 1.0 + 0.01 * rank
    else:
        kernel_ms = tb.solve_batch_timed(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], out,
                                         reps=args.kernel_reps, variant=args.variant)
    # secondary (single-GPU kernel times, not the headline): the reference's full Seidel iteration for
    # every LP (TPR_STRICT_SEIDEL, kernel family 2) and family 2 with its certified shortcuts; the
    # default path must return the same bits as the full iteration
    strict_ms = family2_ms = same_bits = None
    if not args.no_secondary and world == 1:
        reps = max(2, args.kernel_reps // 2)
        full = tb.solve_batch(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], strict=True)
        same_bits = all(bool(torch.equal(torch.nan_to_num(out[k], nan=-7.0), torch.nan_to_num(full[k], nan=-7.0)))
                        for k in ("sd2", "u", "K")) and bool(torch.equal(out["status"], full["status"]))
        strict_ms = tb.solve_batch_timed(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], full,
                                         reps=reps, strict=True)
        family2_ms = tb.solve_batch_timed(dv["coef"], dv["breaks"], dv["grid"], dv["vlim"], dv["alim"], full,
                                          reps=reps, variant=2)
    ok_frac = float((out["status"] == 0).double().mean().item())

    # multi-GPU: what each rank's kernel took and what the gather costs on its own (outside the timed region),
    # so that an N > 1 line explains itself
    per_rank_kernel_ms = gather_alone_ms = None
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=cdev)
        t[rank] = kernel_ms
        dist.all_reduce(t)
        per_rank_kernel_ms = [float(v) for v in t.tolist()]
        g2 = PipelinedGather(B, N + 1, torch.float64, cdev)
        sd2c = out["sd2"].to(cdev)
        g2.submit(sd2c); g2.finish(); sync(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            g2.submit(sd2c)
            g2.finish()
        sync()
        tg = torch.tensor([(time.perf_counter() - t0) / 5 * 1e3], dtype=torch.float64, device=cdev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_alone_ms = float(tg.item())

    if rank == 0:
        bytes_per_traj = algorithmic_bytes(d, N, nseg)
        achieved = bytes_per_traj * B / (kernel_ms * 1e-3) / 1e9
        traj_per_s = world * B * args.steps / elapsed
        pmc = None
        if not stub and world == 1 and not args.no_configs:  # (the full default run; short / profiled runs replay the committed passes)
            pmc = pmc_measure(B, d, N, args.variant, kernel_ms)
        if pmc is None and not stub:
            pmc = pmc_traffic(B, d, N, kernel_ms)
        line = {
            "metric": "trajectories/sec (7-DoF N=200 batch; waypoint-LPs/sec = 3N x this)",
            "value": traj_per_s,
            "unit": "trajectories/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "batch=%d per GPU, %d-DoF random cubic splines (5 waypoints), N=%d gridpoints, "
                            "JointVelocity+JointAcceleration(Interpolation), seidel path, fp64" % (B, d, N),
                "global_batch": world * B, "dof": d, "gridpoints": N,
                "parallelism": ("shard%d+rccl_gather(sd2, %s)" % (world, gather_mode["value"]) if world > 1 else "single")
                               + (" [STUB SOLVER, no GPU: control-flow rehearsal over gloo, timings meaningless]" if stub else
                                  (" [REHEARSAL on one GPU over gloo: timings meaningless]" if args.rehearsal else "")),
                "kernel_variant": args.variant,
            },
            "waypoint_lps_per_s": 3 * N * traj_per_s,
            "gather_placement": (calibration or {"chosen": gather_mode["value"]}) if world > 1 else None,
            "per_rank_kernel_ms": per_rank_kernel_ms,
            "gather_alone_ms": gather_alone_ms,
            "secondary_measured": (not args.no_secondary) and world == 1,
            "full_iteration": {
                "note": "TPR_STRICT_SEIDEL: every stage LP through the reference's full Seidel iteration (kernel "
                        "family 2) instead of the certified answers; single-GPU kernel time only",
                "kernel_ms": strict_ms, "value_per_gpu": (B / strict_ms * 1e3) if strict_ms else None,
                "unit": "trajectories/s",
                "default_path_returns_identical_bits": same_bits,
            },
            "certificates": "trace-following (sound) certificates are the only mode since round 4: a stage LP is answered from a "
                            "certificate only where the reference's whole pivot sequence is predictable (DESIGN.md 3.1); "
                            "TPR_SOUND_CERTIFICATES is accepted and ignored",
            "family2_rows_across_lanes": {
                "note": "kernel family 2 (8 lanes per trajectory) with its certified shortcuts; serves d > 8 and the "
                        "strict mode; single-GPU kernel time only",
                "kernel_ms": family2_ms, "value_per_gpu": (B / family2_ms * 1e3) if family2_ms else None,
                "unit": "trajectories/s",
            },
            "ok_fraction": ok_frac,
            "roofline": {
                # bound / achieved / peak / unit / frac are ONE consistent set: the HBM figures the contract defines --
                # algorithmic bytes per launch / kernel time measured in this run against the 8 TB/s peak (BASELINE's target is
                # quoted against HBM).  What actually BINDS the kernel is fp64 VALU instruction issue: `binding_resource` here
                # and the separate roofline_compute block (no MFMA on this path: there is no dense contraction)
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "hbm_frac": achieved / HBM_PEAK_GBS,
                "binding_resource": "valu_issue",
                "binding_resource_frac": (compute_roofline(pmc, kernel_ms) or {}).get("frac"),
                "traffic": pmc["bytes_per_launch"] if pmc else None,
                "algorithmic_bytes_per_launch": bytes_per_traj * B,
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_trajectory": bytes_per_traj,
                "traffic_source": (pmc["source"] + ("" if pmc.get("measured_in_this_run") else " -- REPLAYED from the committed profile, not measured in this run")) if pmc else None,
                "pmc": pmc,
                "note": "the fused path is bound by fp64 VALU issue and, at one wave per SIMD, by its own dependency "
                        "latencies -- not by HBM: DESIGN.md section 3.8; see roofline_compute",
            },
            "roofline_compute": compute_roofline(pmc, kernel_ms),
        }
        if stub:
            line["stub_gather_delivered_every_ranks_last_step"] = gather_check
        if not args.no_secondary and world == 1:
            line["tolerance_build"] = tolerance_probe(B, d, N, 20240924 + rank, out["sd2"].cpu().numpy(), out["status"].cpu().numpy())
            line["sound_tolerance_build"] = tolerance_probe(B, d, N, 20240924 + rank, out["sd2"].cpu().numpy(), out["status"].cpu().numpy(),
                                                            libname="libtoppra_hip_stol.so", note=NOTE_SOUND_TOLERANCE)
        if not args.no_configs and world == 1:
            line["configs"] = baseline_configs(torch, tb, dev)
            line["end_to_end"] = end_to_end(torch, tb, dev)
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (rank 0's host cores)
            line["cpu_baseline"] = cpu_baseline(data)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""How the host of the GPU box scales the C oracle (bench.py's cpu_baseline): CPU topology / cgroup limits and the
rate at 1 .. all threads.  Host-only; run on the GPU box because that is where the baseline is quoted."""
import os, sys, time
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "spread")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
from toppra_amd import batch as tb
from oracle import oracle as orc
print("logical", os.cpu_count(), "physical", psutil.cpu_count(logical=False), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("loadavg", open("/proc/loadavg").read().strip())
data = tb.make_synthetic_batch(32768, 7, 200)
def rate(n, t):
    t0 = time.perf_counter()
    orc.solve_batch(data["coef"][:n], data["breaks"], data["grid"], data["vlim"][:n], data["alim"][:n], nthreads=t)
    return n / (time.perf_counter() - t0)
rate(512, 1)
r1 = rate(4096, 1)
print("threads 1: %.0f traj/s" % r1)
for t in (2, 4, 8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1):
        break
    rate(64 * t, t)
    r = max(rate(min(32768, 1024 * t), t) for _ in range(2))
    print("threads %d: %.0f traj/s, %.1f x one thread, efficiency %.2f" % (t, r, r / r1, r / r1 / t), flush=True)

"""ANALYSIS TOOL (CPU, test infrastructure): classify the reference's pivot sequences on the backward upper-bound LPs
whose active pair moved (tools/pivot_trace.c, which includes the oracle with its trace hooks).

    python tools/pivot_trace.py [B] [d] [N] [seed]
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toppra_amd import batch  # noqa: E402  (host-side generator only)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20240924
    data = batch.make_synthetic_batch(B, d, N, seed=seed)
    data["vlim"] = data["vlim"] * float(os.environ.get("VSCALE", "1"))
    exe = "/tmp/pivot_trace"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-o", exe,
                           os.path.join(ROOT, "tools", "pivot_trace.c"), "-lm"])
    path = "/tmp/pivot_trace_workload.bin"
    nseg = data["coef"].shape[2]
    with open(path, "wb") as f:
        np.array([B, d, nseg, N], dtype=np.int32).tofile(f)
        for key in ("coef", "breaks", "grid", "vlim", "alim"):
            np.ascontiguousarray(data[key], dtype=np.float64).tofile(f)
    subprocess.check_call([exe, path])


if __name__ == "__main__":
    main()

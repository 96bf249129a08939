#!/usr/bin/env python
"""Every kernel family against the REFERENCE'S OWN compiled solver (oracle/_ref, driven by the reference's passes:
oracle/ref_solver_baseline.py) on the adversarial sliver family of tools/gpu_sliver_hunt.py -- where the reference's run
ends "infeasible" on LPs that have an optimum -- and on scaled natural batches: K, sd, u and the failures, bit for bit.

  python tools/gpu_vs_reference_solver.py [rounds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from toppra_amd import batch  # noqa: E402
from oracle import ref_solver_baseline as rb  # noqa: E402
import gpu_sliver_hunt as hunt  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if not rb.available():
    sys.exit("oracle/_ref holds no compiled reference solver")
total = bad_total = ref_failed = 0
for r in range(rounds):
    for B, d, N, seed in ((768, 7, 60, 101 + 10 * r), (768, 4, 50, 102 + 10 * r), (512, 3, 40, 103 + 10 * r), (512, 8, 48, 104 + 10 * r),
                          (384, 12, 40, 105 + 10 * r), (384, 9, 40, 106 + 10 * r)):
        (coef, breaks, grid, vlim, alim, sd0, sd1), j = hunt.family(B, d, N, seed)
        sd1 = np.round(np.asarray(sd1) * 1024) / 1024  # exact squares: ** in the reference's passes, sd * sd on the device
        ref = []
        for k in range(B):
            vel, acc = rb.constraint_tuples(coef[k], breaks, grid, vlim[k], alim[k])
            w = rb.make_wrapper([rb.PrecomputedConstraint(vel, False), rb.PrecomputedConstraint(acc, True)], None, grid)
            ref.append(rb.parameterization(w, 0.0, float(sd1[k])))
        nfail = sum(1 for x in ref if x[1] is None or np.isnan(x[1]).any())
        line = "B %4d d %2d N %3d seed %3d reference fails %4d :" % (B, d, N, seed, nfail)
        for variant in (0, 2, 3, 4):
            if variant == 3 and d > 13:
                continue
            got = batch.solve_batch(coef, breaks, grid, vlim, alim, sd0, sd1, want_sd=True, variant=variant)
            bad = 0
            for k in range(B):
                sdd, sd, K = ref[k]
                if sd is None:
                    bad += int(got["status"][k] == 0)
                    continue
                same = np.array_equal(got["K"][k], K, equal_nan=True) and np.array_equal(got["sd"][k], sd, equal_nan=True)
                if not np.isnan(sd).any():
                    same = same and np.array_equal(got["u"][k], sdd)
                bad += int(not same)
            line += "  v%d %d differ" % (variant, bad)
            bad_total += bad
        total += B
        ref_failed += nfail
        print(line, flush=True)
print("total %d trajectories x 4 kernel choices against the reference's compiled solver; the reference fails on %d; differing %d" % (total, ref_failed, bad_total))

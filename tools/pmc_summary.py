#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel average of each counter."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
filt = sys.argv[2] if len(sys.argv) > 2 else "solve_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(root + "/prof_pmc*/*counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if filt not in row["Kernel_Name"]:
                continue
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                if k in row:
                    acc[row["Kernel_Name"][:60]]["_" + k] = [float(row[k])]
for kern, ctrs in acc.items():
    print(kern)
    for name in sorted(ctrs):
        v = ctrs[name]
        print("  %-28s %16.1f  (n=%d)" % (name, sum(v) / len(v), len(v)))

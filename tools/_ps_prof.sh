#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for B in ${PS_BATCHES:-256 1536 65536}; do
for dbg in 0 1 2 4 8; do
  rm -rf $OUT/ps_$dbg
  TPR_PS_DEBUG=$dbg TOPPRA_HIP_LIB=$PWD/build_dbg/libtpr_ps.so rocprofv3 --kernel-trace --stats -d $OUT/ps_$dbg -o run --output-format csv -- python tools/gpu_param_spline_probe.py $B > $OUT/ps_$dbg.log 2>&1
  python - <<PY
import csv, glob
for p in glob.glob("$OUT/ps_$dbg/*kernel_stats*.csv"):
    for row in csv.DictReader(open(p)):
        if "pcr" in row["Name"]:
            print("B $B debug $dbg: avg %.1f us (min %.1f) over %s calls" % (float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, row["Calls"]))
PY
done
done

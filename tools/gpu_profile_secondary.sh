#!/bin/bash
# rocprofv3 kernel statistics + counters of every kernel but the headline one (tools/gpu_secondary_kernels.py).
# Outputs: gpurun_out/sec_kernel_stats.csv, gpurun_out/sec_pmc.txt.  usage: tools/gpu_profile_secondary.sh [stats|pmc|all]
set -u
MODE="${1:-all}"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rm -rf $OUT/prof_sec*
if [ "$MODE" = "stats" ] || [ "$MODE" = "all" ]; then
  SEC_REPS=3 rocprofv3 --kernel-trace --stats -d $OUT/prof_sec_stats -o run --output-format csv -- python tools/gpu_secondary_kernels.py > $OUT/sec_stats.log 2>&1
  grep -v amdgpu.ids $OUT/sec_stats.log | tail -16
  find $OUT/prof_sec_stats -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/sec_kernel_stats.csv
  head -30 $OUT/sec_kernel_stats.csv | cut -c1-200
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
  i=0
  for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
              "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT"; do
    i=$((i+1))
    SEC_REPS=1 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/prof_secpmc$i -o run --output-format csv -- python tools/gpu_secondary_kernels.py > $OUT/sec_pmc$i.log 2>&1
  done
  python - <<'PY' > $OUT/sec_pmc.txt
import csv, glob, collections, os
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(root + "/prof_secpmc*/*counter_collection.csv")):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:70]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for f in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                if f in row: acc[k]["_" + f] = [float(row[f])]
for k, c in sorted(acc.items()):
    a = {n: sum(v) / len(v) for n, v in c.items()}
    line = "%-72s" % k
    if "SQ_ACTIVE_INST_VALU" in a and "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"] > 0:
        line += " valu_busy %.3f" % (a["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * a["GRBM_GUI_ACTIVE"] / 8))
    if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU", 0) > 0:
        line += " lanes %.1f" % (a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAIT_ANY" in a and a.get("SQ_WAVE_CYCLES", 0) > 0:
        line += " wait %.2f" % (a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"])
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        line += " hbm_MB %.1f (fetch %.1f write %.1f)" % ((a["FETCH_SIZE"] + a["WRITE_SIZE"]) / 1024, a["FETCH_SIZE"] / 1024, a["WRITE_SIZE"] / 1024)
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "_VGPR_Count", "_Accum_VGPR_Count", "_Scratch_Size", "_LDS_Block_Size", "_Grid_Size"):
        if n in a: line += " %s %.4g" % (n.replace("SQ_INSTS_", "i").lstrip("_"), a[n])
    print(line)
PY
  cat $OUT/sec_pmc.txt | cut -c1-400
fi

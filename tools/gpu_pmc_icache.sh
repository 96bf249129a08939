#!/bin/bash
# One extra rocprofv3 PMC pass: instruction-cache and instruction-fetch counters of the bench workload.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INST_LEVEL[A-Z_]*" | sort -u > $OUT/icache_counters.txt
cat $OUT/icache_counters.txt
i=0
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/prof_ic$i -o run --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs --kernel-reps 2 > $OUT/prof_ic$i.log 2>&1
  tail -2 $OUT/prof_ic$i.log | cut -c1-200
done
python tools/pmc_summary.py $OUT solve_kernel --glob 'prof_ic*' 2>/dev/null || python - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
acc=collections.defaultdict(list)
for f in glob.glob(out+'/prof_ic*/**/*counter_collection*.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cert_solve_kernel' in r.get('Kernel_Name',''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY

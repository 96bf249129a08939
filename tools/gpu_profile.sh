#!/bin/bash
# rocprofv3 passes for the bench workload.  Outputs under gpurun_out/prof_*/ (CSV summaries).
# usage: tools/gpu_profile.sh "<bench args>" [pmc]
set -u
ARGS="${1:---steps 3 --warmup 1 --no-cpu-baseline}"
MODE="${2:-stats}"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$MODE" = "stats" ] || [ "$MODE" = "all" ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o run --output-format csv -- python bench.py $ARGS > $OUT/prof_stats.log 2>&1
  tail -3 $OUT/prof_stats.log
  find $OUT/prof_stats -name "*kernel_stats*.csv" | head -1 | xargs -I{} sh -c 'head -8 {}'
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
  i=0
  for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
              "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/prof_pmc$i -o run --output-format csv -- python bench.py $ARGS > $OUT/prof_pmc$i.log 2>&1
    tail -2 $OUT/prof_pmc$i.log | cut -c1-300
  done
fi
ls -R $OUT | head -40

#!/bin/bash
# One GPU-box visit: tests, smoke, bench, rocprof summary.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu ==" 
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench =="
timeout 900 python bench.py ${BENCH_ARGS:-} 2>&1 | tail -5 | tee gpurun_out/bench.log

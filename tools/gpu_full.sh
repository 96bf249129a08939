#!/bin/bash
# Full GPU visit: tests, smoke, bench (+cpu baseline), rocprof stats + PMC, secondary measurements.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
timeout 600 python tools/gpu_measure.py 2>/dev/null > gpurun_out/measure.json; tail -5 gpurun_out/measure.json
rm -rf gpurun_out/prof_*
bash tools/gpu_profile.sh "--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --kernel-reps 2" all > gpurun_out/profile.log 2>&1
tail -3 gpurun_out/profile.log

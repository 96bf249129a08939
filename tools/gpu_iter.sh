#!/bin/bash
# One development iteration on the GPU box: parity of the product build (full iteration vs default on large
# batches), A/B timing of library builds, cycle breakdown and shortcut hit rates of the instrumented builds.
#   tools/gpu_iter.sh [parity] [ab libA.so libB.so ...]
mkdir -p gpurun_out
OUT=gpurun_out/iter.log
: > $OUT
if [ "$1" = "parity" ]; then
  shift
  python -m pytest tests/test_gpu_fullsize.py -x -q -k "shortcut_is_exact or all_dofs or headline" 2>&1 | tail -5 >> $OUT
fi
if [ -f build_dbg/libtoppra_dbg.so ]; then TPR_DEV_BUILD=1 TOPPRA_HIP_LIB=$PWD/build_dbg/libtoppra_dbg.so python tools/gpu_shortcut_hitrate.py 2>/dev/null >> $OUT; fi
if [ -f build_dbg/libtoppra_tim.so ]; then TOPPRA_HIP_LIB=$PWD/build_dbg/libtoppra_tim.so python tools/gpu_cert_phases.py 2>/dev/null >> $OUT; fi
if [ "$1" = "ab" ]; then shift; tools/gpu_ab.sh "$@" >> $OUT 2>&1; fi
cat $OUT

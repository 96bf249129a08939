#!/bin/bash
# slack experiment: parity (default vs strict) + hit rates + timing for builds with smaller certificate margins
for s in 1e-7 1e-8; do
  echo "== slack $s"
  TOPPRA_HIP_LIB=$PWD/build_dbg/lib_s$s.so python -m pytest tests/test_gpu_fullsize.py -x -q -k "shortcut_is_exact or all_dofs or headline" 2>&1 | tail -2
  TPR_DEV_BUILD=1 TOPPRA_HIP_LIB=$PWD/build_dbg/libdbg_s$s.so python tools/gpu_shortcut_hitrate.py 2>/dev/null | head -2
  TOPPRA_HIP_LIB=$PWD/build_dbg/libdbg_s$s.so python tools/gpu_walk_fail.py 2>/dev/null | head -8
done
tools/gpu_ab.sh toppra_amd/libtoppra_hip.so build_dbg/lib_s1e-7.so build_dbg/lib_s1e-8.so

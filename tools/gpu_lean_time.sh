#!/bin/bash
# Family 3 at 10..13 dof: which per-stage re-reads from global memory pay (profiles/r04_family3_lean_rereads.log).
# Build the two comparison libraries first (CPU, ~1 min each; the object cache keeps the other translation units):
#   TPR_BUILD_CERT_FLAGS_ABOVE_8="-DTPR_LEAN_VEL_FROM=10 -DTPR_LEAN_COEF_FROM=99" python -m toppra_amd.build --out=build/exp/lib_vel.so
#   TPR_BUILD_CERT_FLAGS_ABOVE_8="-DTPR_LEAN_VEL_FROM=99 -DTPR_LEAN_COEF_FROM=10" python -m toppra_amd.build --out=build/exp/lib_coef.so
# then on the GPU box: bash tools/gpu_lean_time.sh
for d in 10 11 12 13; do
  for lib in "" build/exp/lib_vel.so build/exp/lib_coef.so; do
    if [ -n "$lib" ]; then
      [ -f "$lib" ] && TOPPRA_HIP_LIB=$PWD/$lib TPR_TIME_DOF=$d python tools/gpu_time_lib.py 2>&1 | tail -1
    else
      TPR_TIME_DOF=$d python tools/gpu_time_lib.py 2>&1 | tail -1
    fi
  done
done

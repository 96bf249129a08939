for d in 10 11 12 13; do
  for lib in "" build/exp/lib_vel.so build/exp/lib_coef.so; do
    if [ -n "$lib" ]; then TOPPRA_HIP_LIB=$PWD/$lib TPR_TIME_DOF=$d python tools/gpu_time_lib.py 2>&1 | tail -1; else TPR_TIME_DOF=$d python tools/gpu_time_lib.py 2>&1 | tail -1; fi
  done
done

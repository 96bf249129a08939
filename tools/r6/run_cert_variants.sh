#!/bin/bash
# Round-6 forensics, GPU side: link every variant object of build_dbg/r6/<dof>/variants with the common objects and run
# the probe on it.   usage: tools/r6/run_cert_variants.sh <dof> <probe.py> [probe args]
dof=$1; probe=$2; shift 2
dir=build_dbg/r6/$dof
mkdir -p gpurun_out /tmp/r6libs
for o in $dir/variants/*.o; do
  n=$(basename $o .o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/r6libs/$n.so $dir/common/*.o $o 2>/tmp/r6libs/$n.linkerr || { echo "$n LINKFAIL"; continue; }
  res=$(TOPPRA_HIP_LIB=/tmp/r6libs/$n.so timeout 120 python $probe "$@" 2>&1 | tail -1)
  echo "$n $res"
  rm -f /tmp/r6libs/$n.so
done

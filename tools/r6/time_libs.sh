#!/bin/bash
# Round 6: kernel times per dof (tools/r5/time_dofs.py) for every library build_dbg/L*.so, then a parity probe of each
# (tools/r6/inst_probe.py) at the same dofs.   usage: tools/r6/time_libs.sh "9 10 11 12 13"
dofs=${1:-"9 10 11 12 13"}
for l in build_dbg/L*.so; do
  TOPPRA_HIP_LIB=$l python tools/r5/time_dofs.py $dofs 2>&1 | grep -v amdgpu.ids
done
for l in build_dbg/L*.so; do
  for d in $dofs; do echo "$(basename $l) d $d parity: $(TOPPRA_HIP_LIB=$l timeout 300 python tools/r6/inst_probe.py $d 2>&1 | tail -1 | cut -c1-200)"; done
done

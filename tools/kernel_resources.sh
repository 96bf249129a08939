#!/bin/bash
# Register / LDS / spill figures of the shipped code object: tools/kernel_resources.sh [lib.so] [name filter]
LIB=${1:-toppra_amd/libtoppra_hip.so}
FILTER=${2:-solve_kernel}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=/dev/stdout $LIB) --output=$TMP/co.o --unbundle 2>/dev/null \
  || { /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/co.o --unbundle; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/co.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if '$FILTER' in name:
        agpr=blk.split()[0]
        print('%-90s vgpr %s agpr %s sgpr %s spill_vgpr %s scratch %s lds %s' % (name[:90], g('vgpr_count'), agpr, g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
rm -rf $TMP

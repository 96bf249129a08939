#!/bin/bash
# Round-6 forensics: device LLVM IR (from tools/r6/two_stage.sh, KEEP_IR=...) -> host object with the code object embedded.
#   tools/r6/ir2obj.sh <dof> <dev.ll> <out.o> [llc flags ...]
set -e
dof=$1; ir=$2; out=$3; shift 3
L=/opt/rocm/lib/llvm/bin
src=$(dirname $0)/../../toppra_amd/csrc/tpr_cert_tu.hip
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-variable -DTPR_TU_D=$dof"
w=$(mktemp -d)
$L/llc -O3 -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -filetype=obj --relocation-model=pic "$@" $ir -o $w/dev.o 2>/dev/null
$L/ld.lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $w/dev.o -o $w/dev.out
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$w/dev.out -output=$w/dev.hipfb
/opt/rocm/bin/hipcc $BASE --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $w/dev.hipfb -c -o $out $src 2>/dev/null
rm -rf "$w"

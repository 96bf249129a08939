#!/bin/bash
# Round-6 forensics: compile csrc/tpr_cert_tu.hip for one dof in TWO stages -- clang to optimised LLVM IR, then (after an
# optional edit of the IR) llc -> lld -> bundler -> host object -- so that what the IR optimiser and what the code generator
# contribute to a wrong result can be told apart.
#   tools/r6/two_stage.sh <dof> <out.o> <ir-edit: none|strip_tbaa> [clang flags ...]
set -e
dof=$1; out=$2; edit=$3; shift 3
L=/opt/rocm/lib/llvm/bin
src=$(dirname $0)/../../toppra_amd/csrc/tpr_cert_tu.hip
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-variable -DTPR_TU_D=$dof"
w=$(mktemp -d)
/opt/rocm/bin/hipcc $BASE "$@" --cuda-device-only -emit-llvm -S -o $w/dev.ll $src
case $edit in
  strip_tbaa) sed -E 's/, !tbaa(\.struct)? ![0-9]+//g' $w/dev.ll > $w/dev2.ll ;;
  none) cp $w/dev.ll $w/dev2.ll ;;
esac
[ -n "$KEEP_IR" ] && cp $w/dev2.ll $KEEP_IR
$L/llc -O3 -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -filetype=obj --relocation-model=pic $LLC_FLAGS $w/dev2.ll -o $w/dev.o
$L/ld.lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $w/dev.o -o $w/dev.out
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$w/dev.out -output=$w/dev.hipfb
/opt/rocm/bin/hipcc $BASE "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $w/dev.hipfb -c -o $out $src
rm -rf $w

#!/bin/bash
# Round 3's wrong-result class: rebuilds the round-3 library (commit fc4e883) with its SOUND
# instantiations of kernel family 3 enabled above 8 dof -- round 3 shipped them disabled because
# cert_feasible_kernel<11 | 13, Interpolation, sound> returned wrong feasible sets -- in variants that differ in ONE thing:
#   a  as round 3 had it                                           -> wrong results at 11 and 13 dof (reproduces)
#   b  CertStage::fetch (slim blocks) without the conditionally-needed load: `lim[(acc ? (k & 3) : vi) * BS]` merged with
#      the select chain's value under an opaque mask instead of `lim[(acc ? 0 : vi) * BS]` and `acc ? ... : cs` -- the
#      SAME values, a semantically identical source                 -> bit-exact
#   c  a + -mllvm -amdgpu-spill-sgpr-to-vgpr=0  (above 8 dof)       -> wrong
#   d  a + -O1                                   (above 8 dof)       -> bit-exact
#   e  a + -mllvm -amdgpu-opt-vgpr-liverange=0                       -> wrong
#   f  a + -mllvm -amdgpu-opt-exec-mask-pre-ra=0                     -> wrong
#   g  a + -mllvm -amdgpu-prealloc-sgpr-spill-vgprs=1                -> wrong (fewer trajectories)
#   h  a + -mllvm -disable-machine-sink                              -> wrong, and now the FAST solve at 13 dof too
# (profiles/r04_r3_miscompile_variants.log).  CPU only (hipcc cross-compiles); ~2 min per variant.  Then, on a GPU box:
#   for v in a b c d e f g h; do python tools/r3_miscompile/repro.py build/r3_repro/$v; done
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/build/r3_src
rm -rf "$SRC"; mkdir -p "$SRC"
git -C "$ROOT" --work-tree="$SRC" checkout fc4e883 -- include toppra_amd
git -C "$ROOT" reset -q
cd "$SRC"
# sound instantiations above 8 dof: instantiate them, and let the dispatcher use them
sed -i 's/constexpr bool kSoundHere = TPR_TU_D <= 8;/constexpr bool kSoundHere = true;/' toppra_amd/csrc/tpr_cert_tu.hip
sed -i 's/A.d <= ((A.flags \& TPR_SOUND_CERTIFICATES) ? 8 : TPR_CERT_MAX_DOF)/A.d <= TPR_CERT_MAX_DOF/' toppra_amd/csrc/tpr_kernels.hip
python - <<'PY'
p = 'toppra_amd/csrc/tpr_cert.hip.inc'; s = open(p).read()
old = """            const double cs = lim[(acc ? 0 : vi) * BS];
            double cp = 0.0, cn = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) { const bool hit = k == j; cp = hit ? cpos[j] : cp; cn = hit ? cneg[j] : cn; }
            c = acc ? ((blk & 1) ? cn : cp) : cs;
"""
new = """#ifdef TPR_R3_FETCH_FIX
            const double cs = lim[(acc ? (k & 3) : vi) * BS];
            double cp = 0.0, cn = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) { const bool hit = k == j; cp = hit ? cpos[j] : cp; cn = hit ? cneg[j] : cn; }
            const double cv = (blk & 1) ? cn : cp;
            int m = -(int)acc;
            asm volatile("" : "+v"(m));
            const int hs = __double2hiint(cs), ls = __double2loint(cs);
            c = __hiloint2double(hs ^ ((hs ^ __double2hiint(cv)) & m), ls ^ ((ls ^ __double2loint(cv)) & m));
#else
""" + old + "#endif\n"
assert old in s
open(p, 'w').write(s.replace(old, new))
PY
build_variant() {  # name, flags for the translation units above 8 dof
  TPR_BUILD_CERT_FLAGS_ABOVE_8="$2" python -m toppra_amd.build --force > /dev/null
  mkdir -p "$ROOT/build/r3_repro/$1/toppra_amd"
  cp toppra_amd/*.py toppra_amd/libtoppra_hip.so "$ROOT/build/r3_repro/$1/toppra_amd/"
  echo "built $1 ($2)"
}
for v in ${VARIANTS:-a b d}; do
  case $v in
    a) build_variant a "" ;;
    b) build_variant b "-DTPR_R3_FETCH_FIX" ;;
    c) build_variant c "-mllvm -amdgpu-spill-sgpr-to-vgpr=0" ;;
    d) build_variant d "-O1" ;;
    e) build_variant e "-mllvm -amdgpu-opt-vgpr-liverange=0" ;;
    f) build_variant f "-mllvm -amdgpu-opt-exec-mask-pre-ra=0" ;;
    g) build_variant g "-mllvm -amdgpu-prealloc-sgpr-spill-vgprs=1" ;;
    h) build_variant h "-mllvm -disable-machine-sink" ;;
  esac
done

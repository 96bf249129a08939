#!/bin/bash
# Instruction mix of the largest basic blocks of a device function (development aid).
#   tools/bb_mix.sh [mangled-name-substring] [extra hipcc flags...]
NAME=${1:-_ZN3tpr17cert_solve_kernelILi7ELi64ELb0ELb1EEEvNS_9GroupArgsE}
shift
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTPR_CERT_DEV "$@" --cuda-device-only -S -o $TMP/k.s toppra_amd/csrc/tpr_kernels.hip 2>/dev/null
python3 - "$TMP/k.s" "$NAME" <<'PY'
import re,collections,sys
txt=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(txt) if l.startswith(name+":")][0]
lines=[]
for l in txt[start:]:
    if l.startswith('.Lfunc_end'): break
    lines.append(l)
blocks=[];cur=None
for l in lines:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur=[m.group(1),[]];blocks.append(cur);continue
    if cur is None: continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[1].append(t.split()[0])
print("blocks",len(blocks),"instructions",sum(len(b[1]) for b in blocks))
for name,ins in sorted(blocks,key=lambda b:-len(b[1]))[:8]:
    c=collections.Counter(ins)
    g=lambda pre: sum(v for k,v in c.items() if k.startswith(pre))
    print("%-12s %5d valu %5d salu %4d waitcnt %2d nop %3d ds %3d cndmask %4d accvgpr %4d lane-rw %3d mul %3d add %3d fma %3d cmp %3d max %3d" % (
        name,len(ins),g('v_'),g('s_')-c['s_waitcnt']-c['s_nop'],c['s_waitcnt'],c['s_nop'],g('ds_'),sum(v for k,v in c.items() if 'cndmask' in k),
        sum(v for k,v in c.items() if 'accvgpr' in k),c['v_readlane_b32']+c['v_writelane_b32'],c['v_mul_f64'],c['v_add_f64'],c['v_fma_f64']+c['v_fmac_f64_e32'],g('v_cmp'),c['v_max_f64']))
PY
rm -rf $TMP

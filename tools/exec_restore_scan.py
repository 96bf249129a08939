"""Static check for the LLVM AMDGPU defect behind the "compiler sensitivity" incidents of rounds 3 - 5
(profiles/r06_miscompile_root_cause.md).

THE DEFECT.  A divergent region ends in a join block whose first instruction restores the execution mask,
`$exec = S_OR_B64 $exec, <saved mask>` (the lowered SI_END_CF).  Everything the register allocator inserts "at the top" of
such a block must go BELOW that instruction: above it, a vector instruction still runs under the mask of the region that is
being closed (or under an empty mask when the region was skipped), so the lanes outside the region keep a stale register.
`MachineBasicBlock::SkipPHIsLabelsAndDebug` finds the place with `SIInstrInfo::isBasicBlockPrologue`, which does NOT count a
plain SGPR `COPY` as prologue -- and the SGPR allocation phase (AMDGPU allocates SGPRs, then WWM registers, then VGPRs) leaves
exactly such split copies in front of the S_OR_B64 when scalar registers are short.  The VGPR phase's live-range split then
stops at that COPY and puts its own VGPR->AGPR copy ABOVE the exec restore.  Seen in cert_solve_kernel<5, ..., SDFWD> with the
sign-bit slides: `v_accvgpr_write_b32 a0, v228 ; a1, v229` (q''_4 of the current gridpoint) before `s_or_b64 exec, exec, s[0:1]`
in the forward scan's "segment changed?" join block -- every lane whose gridpoint stayed in its spline segment multiplied x by
a stale q''_4.  Input-independent, decided by register pressure, invisible in the source.

TWO MODES.
  --mir FILE...     MIR after the LAST register-allocation phase (`-mllvm -stop-after=virtregrewriter,2`): exact -- every basic
                    block is still its own block.  Flags any instruction that touches a vector register (VGPR / AGPR) above
                    the block's exec-widening instruction.  V_READLANE / V_WRITELANE / SGPR spill pseudos ignore exec: fine.
  --lib LIB.so      the code objects bundled in a built library (llvm-objdump): join blocks are the targets of
                    `s_cbranch_execz`; flags vector / memory instructions between such a target and its `s_or_b64 exec, exec, ..`.
                    Cheap (no compile) and what tests/test_kernel_resources.py runs on the product library; it cannot see a
                    join block whose skip branch was removed (short regions), the MIR mode can.
  --tu DOF [flags]  compile csrc/tpr_cert_tu.hip for DOF to that MIR with the product's flags (+ extra flags) and scan it.
  --product         every translation unit of the product library (toppra_amd/build.py's job list) through the MIR mode: the
                    release check (minutes: one more device compile per unit); log under profiles/.

Exit status 1 when anything is flagged.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toppra_amd.codegen_check import scan_mir, tu_mir  # noqa: E402
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _functions(text):
    """{name: [(addr, mnemonic, operands, target or None)]} from `llvm-objdump -d`."""
    out, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            base, cur = int(m.group(1), 16), m.group(2)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+):", line)
        if cur is None or not m:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        t = re.search(r"<%s\+0x([0-9a-f]+)>" % re.escape(cur), line)
        out[cur].append((addr, mn, ops, base + int(t.group(1), 16) if t else None))
    return out


def scan_code_object(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(elf_bytes)
    try:
        text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
    finally:
        os.unlink(f.name)
    hits = []
    for fn, insts in _functions(text).items():
        joins = {t for (_, mn, _, t) in insts if mn == "s_cbranch_execz" and t is not None}
        index = {a: i for i, (a, _, _, _) in enumerate(insts)}
        for j in sorted(joins):
            i = index.get(j)
            above = []
            while i is not None and i < len(insts):
                a, mn, ops, _ = insts[i]
                if mn.startswith("s_or_b64") and ops.replace(" ", "").startswith("exec,exec,"):
                    bad = [x for x in above if not x.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "s_"))]
                    if bad:
                        hits.append((fn, "+0x%x" % (j - insts[0][0]), bad))
                    break
                if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) or "saveexec" in mn or (mn.startswith("s_") and ops.startswith("exec")):
                    break
                above.append(mn + " " + ops)
                i += 1
    return hits


def scan_lib(lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    from concurrent.futures import ThreadPoolExecutor
    hits = []
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:  # (llvm-objdump does the work: threads are enough)
        for h in pool.map(scan_code_object, kr.code_objects(lib)):
            hits += h
    return hits


def main(argv):
    hits = []
    if argv and argv[0] == "--mir":
        for p in argv[1:]:
            hits += scan_mir(p)
    elif argv and argv[0] == "--lib":
        hits = scan_lib(argv[1] if len(argv) > 1 else os.path.join(ROOT, "toppra_amd", "libtoppra_hip.so"))
    elif argv and argv[0] == "--tu":
        p = tu_mir(int(argv[1]), argv[2:])
        hits = scan_mir(p)
        os.unlink(p)
    elif argv and argv[0] == "--product":
        from concurrent.futures import ThreadPoolExecutor
        sys.path.insert(0, ROOT)
        from toppra_amd import build as B
        # (source, defines, the unit's own flags): family 3's units as build.py compiles them -- per-dof flags, split units
        units = [("tpr_kernels.hip", ["-DTPR_CERT_MAX_DOF=%d" % B.CERT_MAX_DOF], [])]
        for d in B.CERT_DOFS:
            if d in B.CERT_UNIT_PARTS:
                units += [("tpr_cert_tu.hip", ["-DTPR_TU_D=%d" % d, "-DTPR_TU_PART=%d" % part], list(fl)) for part, fl in sorted(B.CERT_UNIT_PARTS[d].items())]
            else:
                units.append(("tpr_cert_tu.hip", ["-DTPR_TU_D=%d" % d], list(B.CERT_UNIT_FLAGS.get(d, []))))
        units += [("tpr_robust_tu.hip", ["-DTPR_TU_HALF=%d" % h], []) for h in (0, 1)] + [("tpr_dense_tu.hip", [], [])]

        def one(u):
            p = tu_mir(0, list(u[2]) + argv[1:], source=u[0], defines=u[1])
            try:
                nblk = sum(1 for l in open(p, errors="replace") if l.startswith("  bb."))
                return u, scan_mir(p), nblk
            finally:
                os.unlink(p)
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:
            for u, h, nblk in pool.map(one, units):
                print("%-20s %-34s %-70s %6d blocks  %d flagged" % (u[0], " ".join(u[1]), " ".join(u[2]), nblk, len(h)), flush=True)
                hits += h
    else:
        print(__doc__)
        return 2
    for fn, blk, bad in hits:
        print("%s  %s" % (fn, blk))
        for b in bad[:6]:
            print("      " + b[:160])
    print("%d block(s) with vector instructions above the exec restore" % len(hits))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

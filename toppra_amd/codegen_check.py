"""Build-time check of the generated code for ONE known LLVM AMDGPU register-allocation defect: a vector copy placed ABOVE a join
block's exec restore (profiles/r06_miscompile_root_cause.md -- the cause of the wrong results of rounds 3 and 5, which depended on
value-identical spellings of unrelated source).

A divergent region ends in a join block whose first instruction restores the execution mask, `$exec = S_OR_B64 $exec, <saved>`
(the lowered SI_END_CF).  What the register allocator inserts "at the top" of such a block must go BELOW that instruction; above
it, a vector instruction still runs under the mask of the region being closed (or an empty mask when the region was skipped) and
the other lanes keep a stale register.  `MachineBasicBlock::SkipPHIsLabelsAndDebug` looks for the place with
`SIInstrInfo::isBasicBlockPrologue`, which does not count a plain SGPR `COPY` as prologue -- and AMDGPU allocates SGPRs first, whose
live-range splits leave exactly such copies in front of the S_OR_B64 when scalar registers are short (these kernels spill 100 - 190
of them).  The VGPR phase's split then stops at that COPY and puts its own copy above the exec restore.

`scan_mir` reads MIR taken after the LAST register-allocation phase (`-mllvm -stop-after=virtregrewriter,2`), where every basic
block is still its own block, and reports each instruction that touches a VGPR / AGPR above a block's exec-widening instruction
(V_READLANE / V_WRITELANE / SGPR spill pseudos ignore exec and are fine).  `toppra_amd.build` runs it on every translation unit of
kernel family 3 (TPR_BUILD_VERIFY=0 turns that off) and refuses to link a library that contains the pattern;
`tools/exec_restore_scan.py` is the command-line front end (+ a disassembly-based variant for built libraries)."""
import os
import re
import subprocess
import sys
import tempfile

EXEC_BLIND = ("V_READLANE", "V_WRITELANE", "V_READFIRSTLANE", "SI_SPILL_S", "SI_RESTORE_S32_FROM_VGPR", "SI_SPILL_S32_TO_VGPR",
              "IMPLICIT_DEF", "KILL", "DBG_", "CFI_", "BUNDLE")


def scan_mir(path):
    """[(function, block, [instructions above the exec restore])]"""
    hits, fn, blk, above, done = [], None, None, [], True
    widen = re.compile(r"\$exec(_lo)? = S_OR_B(64|32) \$exec|S_OR_SAVEEXEC_B(64|32)")
    for line in open(path, errors="replace"):
        s = line.strip()
        if line.startswith("name:"):
            fn = s.split()[-1]
            continue
        m = re.match(r"^  bb\.(\d+)", line)
        if m:
            blk, above, done = s.rstrip(":"), [], False
            continue
        if done or not s or s.startswith(("successors:", "liveins:", ";")) or not line.startswith("    "):
            continue
        head = s.split("::")[0]
        if widen.search(head):
            bad = [x for x in above if re.search(r"\$(vgpr|agpr)\d", x) and not any(k in x for k in EXEC_BLIND)]
            if bad:
                hits.append((fn, blk, bad))
            done = True
            continue
        # any other instruction that defines exec, or a terminator, comes first: not a plain join block
        if re.search(r"(^|\s)\$exec(_lo|_hi)? = |implicit-def (dead )?\$exec|SAVEEXEC|^S_CBRANCH|^S_BRANCH|^S_ENDPGM|^SI_[A-Z_]*(IF|ELSE|LOOP|END_CF)", head):
            done = True
            continue
        above.append(s)
    return hits


def tu_mir(dof, extra=(), source="tpr_cert_tu.hip", defines=None):
    """MIR of one translation unit after the last register-allocation phase, with the product's flags."""
    from . import build as B
    flags = [f for f in B.FLAGS if f not in ("-shared",)]
    defs = defines if defines is not None else ["-DTPR_TU_D=%d" % dof]
    out = tempfile.NamedTemporaryFile(suffix=".mir", delete=False).name
    cmd = [B.hipcc()] + flags + defs + list(extra) + ["--cuda-device-only", "-S", "-mllvm", "-stop-after=virtregrewriter,2", "-o", out,
                                                      os.path.join(B.CSRC, source)]
    subprocess.run(cmd, cwd=B.CSRC, check=True, capture_output=True)
    return out



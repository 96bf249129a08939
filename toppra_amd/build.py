"""Build libtoppra_hip.so (the HIP kernels + C-ABI) in-tree for gfx950.

``python -m toppra_amd.build`` or ``toppra_amd.build.build()``.  hipcc cross-compiles without a
GPU, so this also runs in the CPU-only build container.  ``-ffp-contract=off`` is mandatory: the
parity target is the reference's FMA-free x86-64 arithmetic (see csrc/tpr_device.hpp).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtoppra_hip.so")
SOURCES = ["tpr_kernels.hip", "tpr_cert_tu.hip", "tpr_robust_tu.hip", "tpr_dense_tu.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-Wno-unused-variable"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libtoppra_hip.so)")


def deps():
    out = []
    for name in os.listdir(CSRC):
        if name.endswith((".hip", ".hpp", ".inc", ".h")):
            out.append(os.path.join(CSRC, name))
    out.append(os.path.join(HERE, "..", "include", "toppra_hip.h"))
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in deps())


# kernel family 3: one translation unit per dof (csrc/tpr_cert_tu.hip), 1..14 (slim blocks above 8 dof; 14 dof stores K without
# staging, which keeps its block under 160 KB / 4).  Round 4's
# trace-following certificates first pushed the 9..13-dof instantiations far out of the register file (1.4 - 2.4 KB of scratch
# per lane: 10.6 - 22 ms at 65536 x d x 200); the cause was one conditionally-needed load in CertStage::fetch that the compiler
# sank into divergent regions (tpr_cert_lane.hip.inc), and without it they are back at 0 - 0.7 KB: 3.0 / 4.6 / 7.8 / 7.7 /
# 10.9 ms at 9..13 dof against 10.2 - 12.0 for the rows-across-lanes kernels.
CERT_MAX_DOF = int(os.environ.get("TPR_BUILD_CERT_MAX_DOF", "15"))
CERT_DOFS = tuple(range(1, CERT_MAX_DOF + 1))


# Per-dof compiler flags of kernel family 3's units, chosen by TIMING among the code generations that pass the check of
# codegen_check.py (65536 x d x 200, profiles/r06_dofs_matrix.log, profiles/r06_sched_flags_9_13.log): above 8 dof the kernels fill
# the register file, and what the scheduler and the allocator make of them moves by tens of percent with flags that change nothing
# else.  Round 6, second session: the pre-RA scheduler's direction and its register-pressure trackers matter most -- top-down
# list scheduling takes the 12-dof solve from 4.62 to 3.98 ms and (with -fno-slp-vectorize) the 13-dof one from 8.03 to 5.32 ms
# (scratch per lane 328 -> 132 B); the AMDGPU pressure trackers take 10 dof from 2.97 to 2.80 ms.
_TRACKERS = ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"]
_TOPDOWN = ["-mllvm", "-misched-prera-direction=topdown"]
CERT_UNIT_FLAGS = {8: _TOPDOWN + ["-fno-slp-vectorize"],  # (8 dof: 2.15 / 2.29 / 3.08 -> 2.11 / 2.19 / 2.94 ms solve / feasible sets / TOPPRAsd;
                   # up to 7 dof nothing moves by more than 1 - 2 %: profiles/r06_sched_flags_6_8.log)
                   9: ["-fno-slp-vectorize"] + _TRACKERS, 10: ["-fno-slp-vectorize"] + _TRACKERS, 12: _TOPDOWN,
                   13: _TOPDOWN + ["-fno-slp-vectorize"],
                   14: _TRACKERS}  # (12 .. 14 dof are split by entry point: CERT_UNIT_PARTS below; these are the flags of an unsplit experiment build)
# A dof whose three entry points (1 = fused solve / backward scan, 2 = feasible sets, 3 = TOPPRAsd; csrc/tpr_cert_tu.hip,
# -DTPR_TU_PART) want different flags is compiled as three units: {dof: {part: flags}}.  Timing choices as above.
_MAXILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
_REVERSE = ["-mllvm", "-greedy-reverse-local-assignment=1"]
CERT_UNIT_PARTS = {
    # 7 dof (the headline shape): the solve kernel does not react to any of 28 settings; feasible sets 1.86 -> 1.81 ms with the
    # max-ILP strategy, TOPPRAsd 2.59 -> 2.48 ms per call top-down without SLP (profiles/r06_part_flags.log)
    7: {1: [], 2: _MAXILP, 3: _TOPDOWN + ["-fno-slp-vectorize"]},
    # 12 .. 14 dof (profiles/r06_part_flags_round2.log, on the code with the smaller exchange area): solve / feasible sets / TOPPRAsd
    # 12: 3.74 / 6.86 / 5.87 ms, 13: 4.45 / 8.20 / 6.44 ms, 14: 6.97 / 7.46 / 10.6 ms -- one flag set per unit costs up to 2 x on
    # one of the three (13 dof, TOPPRAsd: 6.4 ms with trackers + reverse assignment, 11.8 top-down + reverse, which is the solve's best)
    12: {1: _TOPDOWN + _REVERSE, 2: _REVERSE, 3: _TOPDOWN},
    13: {1: _TOPDOWN + _REVERSE, 2: _TOPDOWN, 3: _TRACKERS + _REVERSE},
    14: {1: _TRACKERS + ["-fno-slp-vectorize"], 2: _TRACKERS + _REVERSE, 3: _TRACKERS},
    # 15 dof (one 16-lane batch group: 40.5 KB of LDS): 9.6 / 10.3 / 12.1 ms against 13.9 for the rows-across-lanes solve.  As ONE unit
    # this dof's TOPPRAsd kernel came out wrong (the allocator dropped a dword of a split register tuple:
    # profiles/r06_dof15_unsplit_unit_incident.log); all thirty split builds of profiles/r06_dof15_parts.log pass both nets.
    15: {1: _TRACKERS + _REVERSE, 2: _MAXILP, 3: _TRACKERS + ["-fno-slp-vectorize"]},
}
# ... and what the build tries next, in this order, when a unit's code shows a vector copy above an exec restore
# (profiles/r06_miscompile_root_cause.md): the first clean code generation is linked, none is an error.
CERT_FLAG_LADDER = [[], ["-fno-slp-vectorize"], ["-mllvm", "-greedy-reverse-local-assignment=1"],
                    ["-fno-slp-vectorize", "-mllvm", "-greedy-reverse-local-assignment=1"]]


def _compile_and_link(target, flags, defines, verbose, single_tu, cert_max_dof=None):
    """hipcc the translation units in parallel (the certified lane kernels are most of the compile time: one unit per
    dof), then link the objects into `target`.  Instrumented development builds (`defines`) are ONE translation unit:
    their counters are device globals, and they instantiate 7 dof only."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cc = hipcc()
    cflags = [f for f in flags if f != "-shared"]
    dflags = ["-D" + d for d in defines]
    main = os.path.join(CSRC, "tpr_kernels.hip")
    if single_tu:
        cmd = [cc] + flags + dflags + ["-DTPR_SINGLE_TU", "-DTPR_CERT_DEV", "-o", target, main]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        return target
    with tempfile.TemporaryDirectory(prefix="tpr_build_") as tmp:
        max_dof = cert_max_dof or CERT_MAX_DOF
        jobs = [(main, os.path.join(tmp, "main.o"), ["-DTPR_CERT_MAX_DOF=%d" % max_dof])]
        for d in range(1, max_dof + 1):
            extra = os.environ.get("TPR_BUILD_CERT_FLAGS_ABOVE_8", "").split() if d > 8 else []  # (compiler experiments)
            # TPR_BUILD_CERT_FLAGS: experiment flags for family 3's translation units -- with TPR_BUILD_ONLY_CERT_DOFS, for the
            # selected dofs ONLY (the other dofs keep the product's flags, hence the product's cached objects)
            only_dofs = os.environ.get("TPR_BUILD_ONLY_CERT_DOFS", "").split()
            if not only_dofs or str(d) in only_dofs:
                extra += os.environ.get("TPR_BUILD_CERT_FLAGS", "").split()
            parts = CERT_UNIT_PARTS.get(d)
            if parts is None and str(d) in os.environ.get("TPR_BUILD_SPLIT_DOFS", "").split():  # (experiments: split, the unit's flags on every part)
                parts = {k: CERT_UNIT_FLAGS.get(d, []) for k in (1, 2, 3)}
            if not extra and not defines and parts:  # (the product's split unit: one object per entry point)
                for part, pflags in sorted(parts.items()):
                    jobs.append((os.path.join(CSRC, "tpr_cert_tu.hip"), os.path.join(tmp, "cert%dp%d.o" % (d, part)),
                                 ["-DTPR_TU_D=%d" % d, "-DTPR_TU_PART=%d" % part] + list(pflags)))
                continue
            if not extra and not defines:  # (no experiment on this dof: the product's flags for it)
                extra = list(CERT_UNIT_FLAGS.get(d, []))
            jobs.append((os.path.join(CSRC, "tpr_cert_tu.hip"), os.path.join(tmp, "cert%d.o" % d), ["-DTPR_TU_D=%d" % d] + extra))
        for half in (0, 1):  # the robust (conic) kernels: 1..8 dof + the lane kernel, 9..16 dof
            jobs.append((os.path.join(CSRC, "tpr_robust_tu.hip"), os.path.join(tmp, "robust%d.o" % half), ["-DTPR_TU_HALF=%d" % half]))
        jobs.append((os.path.join(CSRC, "tpr_dense_tu.hip"), os.path.join(tmp, "dense.o"), []))  # dense rows: any constraint list
        jobs.sort(key=lambda j: 0 if ("robust" in j[1] or "main" in j[1] or "dense" in j[1]) else 1)  # the longest units first

        # Object cache (git-ignored): an object is reused when its source, every file of csrc/ + the header, its flags AND the
        # compiler (`hipcc --version`) are unchanged.  File names are <unit>_<flags hash>_<sources hash>.o; writes go through a
        # temporary file + os.replace, so that concurrent builders (several ranks, pytest workers) never link a half-written
        # object.  Development shortcut: TPR_BUILD_ONLY_CERT_DOFS="7 12" recompiles only those dofs of kernel family 3 and takes
        # the other dofs' objects from the cache even when their SOURCES are stale -- never with other flags (the tolerance
        # build's -ffp-contract=fast objects carry another flags hash), and never for a release build (__graft_entry__.build()
        # does not set it).
        import hashlib
        cache = os.environ.get("TPR_BUILD_CACHE") or os.path.join(HERE, "..", "build", "objcache")
        if not os.environ.get("TPR_BUILD_CACHE") and (not os.access(os.path.abspath(os.path.join(HERE, "..")), os.W_OK) or "site-packages" in HERE):
            cache = os.path.join(os.path.expanduser("~"), ".cache", "toppra_amd", "objcache")  # an installed package: a user cache
        os.makedirs(cache, exist_ok=True)
        dep_hash = hashlib.sha256()
        for path in sorted(deps()):
            with open(path, "rb") as fh:
                dep_hash.update(fh.read())
        try:
            cc_id = subprocess.run([cc, "--version"], capture_output=True, text=True).stdout
        except OSError:
            cc_id = cc
        only = os.environ.get("TPR_BUILD_ONLY_CERT_DOFS", "").split()
        # (an object in the cache has passed the check: objects are only cached after it)
        verify = os.environ.get("TPR_BUILD_VERIFY", "1") != "0" and not defines

        def run(job):
            src, obj, extra = job
            fkey = hashlib.sha256((cc_id + " ".join(cflags + dflags + extra)).encode()).hexdigest()[:12]
            skey = hashlib.sha256((dep_hash.hexdigest() + os.path.basename(src)).encode()).hexdigest()[:12]
            name = os.path.basename(obj)[:-2]
            cached = os.path.join(cache, "%s_%s_%s.o" % (name, fkey, skey))
            stale_ok = only and name.startswith("cert") and name[4:].split("p")[0] not in only
            if stale_ok and not os.path.exists(cached):
                olds = sorted((f for f in os.listdir(cache) if f.startswith("%s_%s_" % (name, fkey)) and f.endswith(".o")),
                              key=lambda f: os.path.getmtime(os.path.join(cache, f)))
                if olds:
                    cached = os.path.join(cache, olds[-1])
            try:  # (another builder may evict the object between the test and the copy: recompile then)
                if os.path.exists(cached):
                    shutil.copyfile(cached, obj)
                    return obj
            except OSError:
                pass
            # Family 3's units go through the code-generation check (codegen_check.py): the same compile stopped after the last
            # register-allocation phase, scanned for vector copies above an exec restore, in parallel with the real compile.  A
            # unit that shows the pattern is compiled again with the next flags of CERT_FLAG_LADDER; no clean rung: no library.
            rungs = [[]]
            if verify and name.startswith("cert"):
                rungs = [[]] + [r for r in CERT_FLAG_LADDER if r and not all(f in extra for f in r)]
            problems = []
            for rung in rungs:
                cmd = [cc] + cflags + dflags + extra + rung + ["-c", "-o", obj, src]
                if verbose:
                    print(" ".join(cmd))
                checker = None
                if verify and name.startswith("cert"):
                    from concurrent.futures import ThreadPoolExecutor as _TPE
                    from . import codegen_check

                    def check(rung=rung):
                        mir = obj[:-2] + ".mir"
                        subprocess.check_call([cc] + cflags + dflags + extra + rung + ["--cuda-device-only", "-S", "-mllvm", "-stop-after=virtregrewriter,2",
                                                                                       "-o", mir, src], cwd=CSRC, stderr=subprocess.DEVNULL)
                        return codegen_check.scan_mir(mir)
                    checker = _TPE(max_workers=1)
                    pending = checker.submit(check)
                subprocess.check_call(cmd, cwd=CSRC)
                hits = []
                if checker is not None:
                    hits = pending.result()
                    checker.shutdown()
                if not hits:
                    if rung or problems:
                        print("toppra_amd.build: %s %s: clean with %s after %d flagged code generation(s)" % (name, " ".join(extra), rung or "the unit's flags", len(problems)))
                    break
                problems.append("%s: %s" % (" ".join(extra + rung) or "(no extra flags)", "; ".join("%s %s: %s" % (h[0], h[1], h[2][0][:90]) for h in hits)))
            else:
                raise RuntimeError("code-generation check failed for %s with every flag set tried: vector instructions above an exec restore\n  %s\n"
                                   "(profiles/r06_miscompile_root_cause.md; change the unit's spelling -- e.g. TPR_SIGNBITS_*_DOFS -- "
                                   "or set TPR_BUILD_VERIFY=0 for an experiment)" % (os.path.basename(src), "\n  ".join(problems)))
            final = os.path.join(cache, "%s_%s_%s.o" % (name, fkey, skey))
            for f in os.listdir(cache):  # one object per unit and flag set: finished objects only, never another builder's *.tmp
                if f.startswith("%s_%s_" % (name, fkey)) and f.endswith(".o") and f != os.path.basename(final):
                    try:
                        os.remove(os.path.join(cache, f))
                    except OSError:
                        pass
            try:
                tmp_obj = "%s.%d.tmp" % (final, os.getpid())
                shutil.copyfile(obj, tmp_obj)
                os.replace(tmp_obj, final)
            except OSError:
                pass  # (the cache is an optimisation; the object in `obj` is what gets linked)
            return obj

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(run, jobs))
        if os.environ.get("TPR_BUILD_KEEP_OBJS"):  # (tools/r6/build_cert_variants.py: the objects, for linking variants elsewhere)
            for o in objs:
                shutil.copyfile(o, os.path.join(os.environ["TPR_BUILD_KEEP_OBJS"], os.path.basename(o)))
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return target


def build_tolerance(out=None, verbose=False):
    """The opt-in measurement build "what does bit-exactness cost" (DESIGN.md): same sources with
    -DTPR_TOLERANCE_MODE (certified vertices returned as they are, no replication of the reference's
    last-pivot arithmetic), contracted multiply-adds and reciprocal-based division.  Results agree with the
    product to ~1e-12, status codes identical; it is NOT the product library and nothing loads it by default."""
    target = os.path.abspath(out) if out else os.path.join(HERE, "libtoppra_hip_tol.so")
    flags = [f for f in FLAGS if f != "-ffp-contract=off"] + ["-ffp-contract=fast", "-freciprocal-math"]
    # (family 3 up to 8 dof is all the measurement needs: half of the product library's translation units)
    return _compile_and_link(target, flags, ["TPR_TOLERANCE_MODE"], verbose, single_tu=False, cert_max_dof=min(CERT_MAX_DOF, 8))


def build_sound_tolerance(out=None, verbose=False):
    """The second opt-in measurement build (round 5): the product's SOUND, trace-following certificates with tolerance
    arithmetic in what they return (-DTPR_SOUND_TOLERANCE: the verified vertex from a reciprocal estimate instead of the
    reference's last-pivot formulas and their cross-product guard, reciprocal-based quotients in the forward 1-variable LP) --
    same compiler flags as the product, so the cooperative batches' full iteration stays the reference's arithmetic.  It
    separates what SOUNDNESS costs (kept) from what BIT-EXACTNESS costs (dropped); bench.py reports it beside
    `tolerance_build`.  NOT the product library; nothing loads it by default."""
    target = os.path.abspath(out) if out else os.path.join(HERE, "libtoppra_hip_stol.so")
    return _compile_and_link(target, FLAGS, ["TPR_SOUND_TOLERANCE"], verbose, single_tu=False, cert_max_dof=min(CERT_MAX_DOF, 8))


def build(force=False, verbose=False, defines=(), out=None):
    """Build the library.  ``defines`` / ``out`` produce an instrumented copy next to the product one
    (e.g. defines=("TPR_CERT_TIMING", "TPR_CERT_DEV"), used by tools/gpu_cert_phases.py via
    TOPPRA_HIP_LIB); the product library is always built without defines."""
    target = os.path.abspath(out) if out else LIB
    if not defines and not out and not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(target), exist_ok=True)
    return _compile_and_link(target, FLAGS, list(defines), verbose, single_tu=bool(defines))


def ensure_built(verbose=False):
    """Build the product library only if it is missing (bench.py / smoke() safety net on a box that received
    the sources without the in-tree .so; hipcc is part of the ROCm image).  Never rebuilds an existing one:
    a snapshot's file times say nothing about staleness."""
    if not os.path.exists(LIB):
        build(force=True, verbose=verbose)
    return LIB


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--tolerance" in args:
        print(build_tolerance(verbose=True))
        sys.exit(0)
    if "--sound-tolerance" in args:
        print(build_sound_tolerance(verbose=True))
        sys.exit(0)
    defs = [a[2:] for a in args if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in args if a.startswith("--out=")]
    print(build(force="--force" in args, verbose=True, defines=defs, out=outs[0] if outs else None))

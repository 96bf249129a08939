"""Round-6: variants of ONE entry point (part 1 = solve, 2 = feasible sets, 3 = TOPPRAsd) of one dof's unit of kernel family 3, the
other two parts and the rest of the library from the product build (build.py, TPR_BUILD_SPLIT_DOFS).

    python tools/r6/build_part_variants.py <dof> <part> <spec file>     spec: one "name: flags ..." per line
Objects: build_dbg/r6/<dof>p<part>/{common/*.o, variants/<name>.o}; run with tools/r6/run_cert_variants.sh <dof>p<part> <probe> ...
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def main():
    dof, part = int(sys.argv[1]), int(sys.argv[2])
    spec = [l.strip() for l in open(sys.argv[3]) if l.strip() and not l.startswith("#")]
    out = os.path.join(ROOT, "build_dbg", "r6", "%dp%d" % (dof, part))
    common = os.path.join(out, "common"); var = os.path.join(out, "variants")
    os.makedirs(common, exist_ok=True); os.makedirs(var, exist_ok=True)
    os.environ["TPR_BUILD_SPLIT_DOFS"] = str(dof)
    from toppra_amd import build as B
    if not os.listdir(common):
        os.environ["TPR_BUILD_KEEP_OBJS"] = common
        B.build(out=os.path.join(out, "product.so"), verbose=False)
        os.remove(os.path.join(common, "cert%dp%d.o" % (dof, part)))
        os.remove(os.path.join(out, "product.so"))
    cc = B.hipcc()
    base = [f for f in B.FLAGS if f != "-shared"] + ["-DTPR_TU_D=%d" % dof, "-DTPR_TU_PART=%d" % part]
    def one(line):
        name, flags = line.split(":", 1)
        obj = os.path.join(var, name.strip() + ".o")
        if os.path.exists(obj):
            return name, 0
        r = subprocess.run([cc] + base + flags.split() + ["-c", "-o", obj, os.path.join(B.CSRC, "tpr_cert_tu.hip")], cwd=B.CSRC, capture_output=True, text=True)
        if r.returncode:
            open(obj + ".err", "w").write(r.stderr[-4000:])
        return name, r.returncode
    with ThreadPoolExecutor(max_workers=os.cpu_count()) as pool:
        for name, rc in pool.map(one, spec):
            print(name, "ok" if rc == 0 else "COMPILE FAILED")

if __name__ == "__main__":
    main()

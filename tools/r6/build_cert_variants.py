"""Round-6 forensics: build many variants of ONE dof's translation unit of kernel family 3 (csrc/tpr_cert_tu.hip) as bare
object files, plus the common objects of a library, so that a GPU box links and runs them (tools/r6/run_cert_variants.sh).

    python tools/r6/build_cert_variants.py <dof> <spec file>      spec: one "name: flags ..." per line

Objects go to build_dbg/r6/<dof>/{common/*.o, variants/<name>.o} (git-ignored; they travel with gpurun).
"""
import os, subprocess, sys, shutil
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toppra_amd import build as B

def main():
    dof = int(sys.argv[1])
    spec = [l.strip() for l in open(sys.argv[2]) if l.strip() and not l.startswith("#")]
    out = os.path.join(ROOT, "build_dbg", "r6", str(dof))
    common = os.path.join(out, "common"); var = os.path.join(out, "variants")
    os.makedirs(common, exist_ok=True); os.makedirs(var, exist_ok=True)
    if not os.listdir(common):
        # the product's objects of every other unit (cert units up to 8 dof; the library dispatches 9..13 to family 2 then)
        os.environ["TPR_BUILD_CERT_MAX_DOF"] = "8" if dof <= 8 else str(max(13, dof))
        os.environ["TPR_BUILD_KEEP_OBJS"] = common
        import importlib; importlib.reload(B)
        B.build(out=os.path.join(out, "product.so"), verbose=False)
        os.remove(os.path.join(common, "cert%d.o" % dof))
    cc = B.hipcc()
    base = [f for f in B.FLAGS if f != "-shared"] + ["-DTPR_TU_D=%d" % dof]
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

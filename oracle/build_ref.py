"""Build the reference's two Cython modules into ``oracle/_ref`` (TEST INFRASTRUCTURE ONLY).

This compiles the *reference's own* sources where they lie under ``/root/reference``:

* ``toppra/_CythonUtils.pyx``                          (velocity-bound builder, unmodified)
* ``toppra/solverwrapper/cy_seidel_solverwrapper.pyx`` (Seidel LP + seidelWrapper)

The only change is the mechanical numpy-2 fix ``ctypedef np.int_t INT_t`` -> ``ctypedef long INT_t``
(SURVEY.md section 8(c)); it is applied to a scratch copy in a temporary directory, never to the
reference tree and never to this repository.  Only the two built ``.so`` files land in
``oracle/_ref/`` (git-ignored).  Flags follow the reference's ``setup.py:43-52`` (``-O1``).

The built modules exist to (1) pin the C restatement in ``oracle/seidel_oracle.c`` and (2) generate the
golden vectors committed under ``tests/golden``.  Nothing in the product path may import them.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

REF = os.environ.get("TOPPRA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

MODULES = {
    "toppra._CythonUtils": "toppra/_CythonUtils.pyx",
    "toppra.solverwrapper.cy_seidel_solverwrapper": "toppra/solverwrapper/cy_seidel_solverwrapper.pyx",
}


def so_path(modname: str) -> str:
    return os.path.join(OUT, modname + ".so")


def have_reference() -> bool:
    return all(os.path.exists(os.path.join(REF, p)) for p in MODULES.values())


def build(force: bool = False) -> bool:
    """Returns True when both reference modules are built (or were already)."""
    if not have_reference():
        return all(os.path.exists(so_path(m)) for m in MODULES)
    os.makedirs(OUT, exist_ok=True)
    import numpy as np

    inc_py = sysconfig.get_paths()["include"]
    inc_np = np.get_include()
    for modname, rel in MODULES.items():
        dst = so_path(modname)
        src = os.path.join(REF, rel)
        if not force and os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
            continue
        with tempfile.TemporaryDirectory(prefix="toppra_ref_build_") as tmp:
            # keep the package-relative layout so Cython resolves the module's qualified name
            pkgdir = os.path.join(tmp, os.path.dirname(rel))
            os.makedirs(pkgdir)
            d = tmp
            for part in os.path.dirname(rel).split("/"):
                d = os.path.join(d, part)
                open(os.path.join(d, "__init__.py"), "w").close()
            pyx = os.path.join(tmp, rel)
            with open(src, "r", encoding="utf-8") as fh:
                text = fh.read()
            text = text.replace("ctypedef np.int_t INT_t", "ctypedef long INT_t")
            with open(pyx, "w", encoding="utf-8") as fh:
                fh.write(text)
            subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx], cwd=tmp)
            csrc = pyx[:-4] + ".c"
            subprocess.check_call(
                ["gcc", "-O1", "-shared", "-fPIC", "-fwrapv", "-fno-strict-aliasing", "-w",
                 "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                 "-I", inc_py, "-I", inc_np, csrc, "-o", dst, "-lm"])
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("reference modules built:" if ok else "reference not available:", OUT)

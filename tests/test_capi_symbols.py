"""The C-ABI library loads on a CPU-only machine, exports every symbol the header declares, and
refuses to compute without a GPU (no silent fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "toppra_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from toppra_amd import _capi, build
    build.build()
    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(_capi.EXPORTS) == declared
    assert b"gfx950" in lib.tpr_version()


def test_struct_layout_matches_header():
    from toppra_amd import _capi
    # 6 int32 then 8 pointers (the last one: the wrapper object's warm-start state, r3); 5 pointers
    assert ctypes.sizeof(_capi.tpr_problem) == 6 * 4 + 8 * 8
    assert ctypes.sizeof(_capi.tpr_result) == 5 * 8
    assert _capi.tpr_problem.coef.offset == 24
    assert _capi.tpr_problem.active.offset == 24 + 7 * 8
    # the header declares the same members in the same order
    hdr = open(os.path.join(ROOT, "include", "toppra_hip.h")).read()
    body = hdr[hdr.index("typedef struct tpr_problem {"):hdr.index("} tpr_problem;")]
    order = [body.index(name) for name in ("B, d, nseg, N", "flags;", "variant;", "*coef;", "*breaks;", "*grid;", "*vlim;",
                                           "*alim;", "*sd_start;", "*sd_end;", "*active;")]
    assert order == sorted(order)


def test_no_cpu_fallback():
    from toppra_amd import _capi, batch
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    data = batch.make_synthetic_batch(4, 3, 10)
    with pytest.raises(_capi.ToppraHipError):
        batch.solve_batch(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    lib = _capi.load()
    # compute entries refuse to run before a successful tpr_init
    p, keep = _capi.make_problem(data["coef"], data["breaks"], data["grid"], data["vlim"], data["alim"])
    K = np.zeros((4, 11, 2))
    r = _capi.tpr_result(K=K.ctypes.data)
    assert lib.tpr_solve_batch(ctypes.byref(p), ctypes.byref(r), None) < 0
    assert b"tpr_init" in lib.tpr_last_error()


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ anywhere."""
    pkg = os.path.join(ROOT, "toppra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("# oracle", ""), os.path.join(dirpath, f)


def test_integration_stub_declares_the_same_structs():
    """The ctypes stub INTEGRATION.md hands to a toppra maintainer must declare tpr_problem / tpr_result / tpr_dense_problem field for field
    as the package's own binding does (a struct that is one pointer short makes the library read past its end)."""
    from toppra_amd import _capi
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, cls in (("tpr_problem", _capi.tpr_problem), ("tpr_result", _capi.tpr_result),
                      ("tpr_dense_problem", _capi.tpr_dense_problem)):
        block = re.search(r"class %s\(C\.Structure\):\s*_fields_ = \[(.*?)\]\s*(#[^\n]*)?\n\n" % name, text, re.S)
        assert block, name
        fields = re.findall(r'\("(\w+)",\s*C\.(\w+)\)', block.group(1))
        assert [(n, getattr(ctypes, t)) for n, t in fields] == [(n, t) for n, t in cls._fields_], name

"""Import the real reference package (``/root/reference/toppra``) with the Cython modules that
``oracle/build_ref.py`` compiled into ``oracle/_ref``.  TEST INFRASTRUCTURE ONLY.

Works only in the build container (``/root/reference`` does not exist on the GPU box); callers
must treat ``load() is None`` as "reference unavailable" and skip.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

from . import build_ref


class _RefSoFinder(importlib.abc.MetaPathFinder):
    """Resolve the two compiled submodules to the files under oracle/_ref."""

    def find_spec(self, name, path, target=None):
        if name in build_ref.MODULES:
            p = build_ref.so_path(name)
            if os.path.exists(p):
                return importlib.util.spec_from_file_location(name, p)
        return None


_cached = None


def load():
    """Returns the imported reference ``toppra`` module, or None when it cannot be provided."""
    global _cached
    if _cached is not None:
        return _cached
    if not build_ref.have_reference():
        return None
    try:
        build_ref.build()
    except Exception:  # pragma: no cover - toolchain problem
        return None
    os.environ.setdefault("MPLBACKEND", "Agg")
    if not any(isinstance(f, _RefSoFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _RefSoFinder())
    if build_ref.REF not in sys.path:
        sys.path.insert(0, build_ref.REF)
    try:
        _cached = importlib.import_module("toppra")
    except Exception:
        return None
    return _cached

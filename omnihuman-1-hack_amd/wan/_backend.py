"""Locates the kernel binding package whichever way ``wan`` was imported:
as ``omnihuman-1-hack_amd.wan`` (repo root on sys.path) or as top-level
``wan`` (the package directory on sys.path, as the reference's scripts
expect)."""
import importlib
import importlib.util
import os
import sys

_PKG = "omnihuman-1-hack_amd"


def _load():
    if _PKG in sys.modules:
        return sys.modules[_PKG]
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(_PKG, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_PKG] = mod
    spec.loader.exec_module(mod)
    return mod


backend = _load()
ops = importlib.import_module(_PKG + ".ops")

"""Loader for the package directory ``nonuniformffts.jl_amd`` (its name contains a dot, so a plain
``import`` statement cannot reach it).  ``from nufft_pkg import nufft`` gives the package."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "nonuniformffts.jl_amd")
_NAME = "nonuniformffts_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(_NAME, None)
        raise
    return mod


def __getattr__(name):
    if name == "nufft":
        return load()
    raise AttributeError(name)

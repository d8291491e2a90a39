"""Loader for the product package, whose directory name `qpcontrol.jl_amd/` (fixed by the build
contract) is not a valid Python identifier.  `load()` registers it as module `qpcontrol_jl_amd`."""
import importlib.util
import os
import sys

_NAME = "qpcontrol_jl_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "qpcontrol.jl_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod

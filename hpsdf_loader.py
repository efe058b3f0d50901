"""Imports the package directory (whose prescribed name is not a Python identifier) as ``hpsdf_amd``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hp-adaptive-signed-distance-field-octree_amd")


def load():
    if "hpsdf_amd" in sys.modules:
        return sys.modules["hpsdf_amd"]
    spec = importlib.util.spec_from_file_location("hpsdf_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hpsdf_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

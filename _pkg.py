"""Loader for the `tinyllama.cpp_amd/` package directory.

The directory name (mandated by the repo layout) contains a dot, so `import tinyllama.cpp_amd` cannot
work; this helper registers it under the importable alias `tinyllama_cpp_amd`.
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
ALIAS = "tinyllama_cpp_amd"


def load_package():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    pkg_dir = os.path.join(ROOT, "tinyllama.cpp_amd")
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod

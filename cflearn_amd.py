"""Alias loader: `import cflearn_amd` -> the package in ./carefree-learn_amd/ (whose directory
name, fixed by the project layout, contains a hyphen and cannot be imported directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "carefree-learn_amd")
_spec = importlib.util.spec_from_file_location(
    "cflearn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cflearn_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: the package directory is ``pos-evolution_amd/`` (not an importable
name), so ``import pos_evolution_amd`` resolves to it through this loader."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pos-evolution_amd")
_spec = importlib.util.spec_from_file_location(
    "pos_evolution_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pos_evolution_amd"] = _mod
_spec.loader.exec_module(_mod)

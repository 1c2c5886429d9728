"""Import alias: the package directory is named ``sfm-toy-library_amd`` (not a valid Python
identifier), so ``import sfm_toy_library_amd`` is routed to it from here."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sfm-toy-library_amd")
_spec = importlib.util.spec_from_file_location(
    "sfm_toy_library_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sfm_toy_library_amd"] = _mod
_spec.loader.exec_module(_mod)

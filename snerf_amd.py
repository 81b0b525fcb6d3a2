"""Import shim: the package directory is named ``semantic-nerf-for-satellite-data_amd`` (not a valid
Python identifier), so ``import snerf_amd`` loads that directory as the package ``snerf_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantic-nerf-for-satellite-data_amd")
_spec = importlib.util.spec_from_file_location(
    "snerf_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["snerf_amd"] = _mod
_spec.loader.exec_module(_mod)

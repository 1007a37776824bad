"""Import alias: ``import faoctasr`` loads the package kept in the directory
``frequency-aware-inverse-consistent-octa-super-resolution_amd/`` (whose name is not a valid
Python identifier) and registers it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frequency-aware-inverse-consistent-octa-super-resolution_amd")
_spec = importlib.util.spec_from_file_location("faoctasr", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["faoctasr"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: loads the package directory ``gif-synthesis-with-discrete-diffusion_amd/`` (not a valid
Python identifier) under the module name ``gsdd_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gif-synthesis-with-discrete-diffusion_amd")
_spec = importlib.util.spec_from_file_location("gsdd_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gsdd_amd"] = _mod
_spec.loader.exec_module(_mod)

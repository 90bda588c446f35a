"""Import shim: the package directory is named `fhe.rs_amd` (a dot is not importable), so this
module loads it under the name `fhe_rs_amd`.  `import fhe_rs_amd as fhe` gives the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fhe.rs_amd")
_spec = importlib.util.spec_from_file_location(
    "fhe_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fhe_rs_amd"] = _mod
_spec.loader.exec_module(_mod)

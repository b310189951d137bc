"""Import shim: makes the package directory `rust-exp_amd/` (hyphenated, after the reference repo's
name) importable as `rust_exp_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rust-exp_amd")
_spec = importlib.util.spec_from_file_location(
    "rust_exp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rust_exp_amd"] = _mod
_spec.loader.exec_module(_mod)

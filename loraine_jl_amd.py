"""Import alias: the package directory is `loraine.jl_amd/` (a dot is not importable as a
Python identifier), so `import loraine_jl_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loraine.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "loraine_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["loraine_jl_amd"] = _mod
_spec.loader.exec_module(_mod)

"""Import shim: the product package lives in `draco-oxide_amd/` (hyphenated, as the project layout
prescribes), which Python cannot import by name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location(
    "draco_oxide_amd", os.path.join(_here, "draco-oxide_amd", "__init__.py"),
    submodule_search_locations=[os.path.join(_here, "draco-oxide_amd")])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["draco_oxide_amd"] = _mod
_spec.loader.exec_module(_mod)

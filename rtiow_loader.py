"""Import the package directory `raytracing-in-one-weekend_amd/` (not a valid identifier) as `rtiow_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "rtiow_amd" in sys.modules:
        return sys.modules["rtiow_amd"]
    pkg_dir = os.path.join(_ROOT, "raytracing-in-one-weekend_amd")
    spec = importlib.util.spec_from_file_location("rtiow_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rtiow_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

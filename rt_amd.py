"""Import shim: the package directory is named `ray-tracer-engine_amd` (not a
valid Python identifier), so load it under the module name
`ray_tracer_engine_amd`.  Usage: `import rt_amd; rt = rt_amd.load()`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "ray-tracer-engine_amd")
_NAME = "ray_tracer_engine_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod

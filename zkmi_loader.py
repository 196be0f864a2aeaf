"""Import helper: the package directory is `zk-apps_amd/` (hyphenated by the
repo layout contract), so load it under the module name `zk_apps_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    if "zk_apps_amd" in sys.modules:
        return sys.modules["zk_apps_amd"]
    init = os.path.join(_ROOT, "zk-apps_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        "zk_apps_amd", init, submodule_search_locations=[os.path.dirname(init)]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["zk_apps_amd"] = mod
    spec.loader.exec_module(mod)
    return mod

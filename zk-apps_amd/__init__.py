"""zkmi — Python driver (ctypes) over the C ABI in include/zkmi.h.

This package is test/bench plumbing only: every computation happens inside
libzkmi.so (hand-written HIP for gfx950 + host C++).  There is NO CPU fallback:
if the shared library or a GPU is missing, calls raise.

The directory name contains a hyphen (the repo layout contract), so import it
through `load_pkg()` in `zkmi_loader.py` at the repo root, or via
`importlib` under the module name `zk_apps_amd`.
"""
from .binding import (  # noqa: F401
    Zkmi,
    ZkmiError,
    Context,
    lib_path,
    PHASES,
    ERR,
)

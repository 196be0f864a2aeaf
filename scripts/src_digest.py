#!/usr/bin/env python3
"""sha256 (16 hex digits) over the kernel / C-ABI sources -- zk-apps_amd/csrc/*.{hip,hpp,h,Makefile} and include/*.h, file
names included, in sorted order.  ONE definition of the digest: the Makefile compiles it into the library (zkmi_version()
returns it), bench.py / __graft_entry__.smoke() / scripts/pmc_summary.py recompute it from the files beside the library
they loaded and report `library_matches_sources`."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_digest(root=ROOT):
    h = hashlib.sha256()
    for d in ("zk-apps_amd/csrc", "include"):
        base = os.path.join(root, d)
        for name in sorted(os.listdir(base)):
            path = os.path.join(base, name)
            if os.path.isfile(path) and name.endswith((".hip", ".hpp", ".h", "Makefile")):
                h.update(name.encode() + b"\0")
                h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(csrc_digest())

#!/usr/bin/env python3
"""Register / scratch / LDS allocation of every kernel in a built library, read from the code objects themselves.

    python scripts/kernel_resources.py [zk-apps_amd/libzkmi.so] [name-substring ...]

Walks the clang offload bundles embedded in the shared object (section .hip_fatbin), takes each gfx950 code object
(an ELF), and decodes the AMDGPU metadata note (msgpack: amdhsa.kernels).  No GPU and no ROCm tool needed, so the CPU test
suite can hold the allocations the design depends on (tests/test_cpu_host.py::test_kernel_register_budgets): the
accumulation kernels and the quad-split reduction kernels must stay at <= 168 VGPRs without scratch -- three waves per
SIMD, placed beside each other -- and nothing tells you the day an edit loses that except this.
"""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob, arch="gfx950"):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24 : p + 24 + tlen].decode()
            p += 24 + tlen
            if triple.startswith("hip") and triple.endswith(arch) and size:
                yield blob[i + off : i + off + size]
        pos = i + len(MAGIC)


def notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "64-bit ELF expected"
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
        if sh[1] != 7:  # SHT_NOTE
            continue
        off, size = sh[4], sh[5]
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p : p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p : p + descsz]
            p += (descsz + 3) & ~3
            yield name.rstrip(b"\0"), ntype, desc


def kernels(path, arch="gfx950"):
    """{mangled kernel name: dict(vgpr, agpr, sgpr, scratch, lds, wavefront, max_wg)} over all code objects of `path`."""
    blob = open(path, "rb").read()
    out = {}
    for co in code_objects(blob, arch):
        for name, ntype, desc in notes(co):
            if name != b"AMDGPU" or ntype != 32:
                continue
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out[k[".name"]] = dict(
                    vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                    scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0),
                    vgpr_spill=k.get(".vgpr_spill_count", 0), max_wg=k.get(".max_flat_workgroup_size", 0),
                    dynamic_stack=bool(k.get(".uses_dynamic_stack", False)))
    return out


def short_name(mangled):
    """k_accum_g1_nc<Fq28,3,1,...> style label from an Itanium-mangled kernel name (no demangler needed)."""
    import re

    m = re.search(r"\d+(k_[A-Za-z0-9_]+?)(I.*)?$", mangled)
    if not m:
        return mangled
    base, rest = m.group(1), m.group(2) or ""
    field = "BnFq28" if "BnFq28Params" in rest else "Fq2" if "Fq2T" in rest else "Fq28" if "Fq28Params" in rest else \
        "Fr28" if "Fr28Params" in rest else "BnFr28" if "BnFr28Params" in rest else ""
    # integer / bool template arguments up to the parameter list (E...v), e.g. Li3ELi1ELb0ELb0E
    head = rest.split("Ev", 1)[0]
    ints = re.findall(r"L[ijb](\d+)E", head)
    args = ",".join(([field] if field else []) + ints)
    return base + ("<" + args + ">" if args else "")


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "zk-apps_amd/libzkmi.so"
    want = sys.argv[2:]
    ks = kernels(path)
    print("%5s %5s %7s %7s  %s" % ("vgpr", "agpr", "scratch", "lds", "kernel"))
    for name in sorted(ks, key=short_name):
        if want and not any(w in name for w in want):
            continue
        k = ks[name]
        print("%5d %5d %7d %7d  %s" % (k["vgpr"], k["agpr"], k["scratch"], k["lds"], short_name(name)))


if __name__ == "__main__":
    main()

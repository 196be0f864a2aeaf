#!/usr/bin/env python3
"""Writes the relation and a satisfying assignment for integration/ark_fixture (no GPU needed: host code only).

    python scripts/export_relation_for_arkworks.py OUTDIR [log_n]

relation.bin = n_vars, n_pub, n_constraints (u32 LE) then, for A, B, C: rowptr (nc + 1 x u32), col (nnz x u32),
val (nnz x 32-byte LE canonical Fr);  witness.bin = n_vars x 32-byte LE.  Relation: update_note (withdraw) with
Poseidon-5 at N = 2^log_n (default 14, BASELINE config 0's size).  Without an explicit log_n the relation's natural
size is written as well, to OUTDIR + "_2p13" (N = 2^13: the Poseidon relation proper with next to no padding), so that
one run of the fixture pins the relation's constraint system and the prover together
(tests/test_gpu_parity.py::test_arkworks_fixture_if_present takes both directories)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkmi_loader import load_pkg  # noqa: E402


def export(zk, out, lg):
    from test_cpu_host import _note_update_case

    os.makedirs(out, exist_ok=True)
    r1 = zk.update_note_r1cs(lg, 1)
    inp, _ = _note_update_case(zk, 20260, 1)
    wit, _, rc = zk.update_note_witness(lg, 1, inp)
    assert rc == 0 and r1.is_satisfied(wit)
    blob = struct.pack("<III", r1.n_vars, r1.n_pub, r1.n_constraints)
    for m in range(3):
        rp, cl, vl = r1.export(m)
        blob += struct.pack("<%dI" % len(rp), *rp) + struct.pack("<%dI" % len(cl), *cl) + vl
    open(os.path.join(out, "relation.bin"), "wb").write(blob)
    open(os.path.join(out, "witness.bin"), "wb").write(wit)
    print("wrote relation.bin (%d bytes, N = 2^%d, %d constraints) and witness.bin to %s" % (len(blob), lg, r1.n_constraints, out))
    r1.free()


def main():
    out = sys.argv[1]
    zk = load_pkg().Zkmi()
    if len(sys.argv) > 2:
        export(zk, out, int(sys.argv[2]))
    else:
        export(zk, out, 14)
        export(zk, out.rstrip("/") + "_2p13", 13)


if __name__ == "__main__":
    main()

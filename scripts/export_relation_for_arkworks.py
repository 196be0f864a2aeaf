#!/usr/bin/env python3
"""Writes the relation and a satisfying assignment for integration/ark_fixture (no GPU needed: host code only).

    python scripts/export_relation_for_arkworks.py OUTDIR [log_n]

relation.bin = n_vars, n_pub, n_constraints (u32 LE) then, for A, B, C: rowptr (nc + 1 x u32), col (nnz x u32),
val (nnz x 32-byte LE canonical Fr);  witness.bin = n_vars x 32-byte LE.  Relation: update_note (withdraw) with
Poseidon-5 at N = 2^log_n (default 14, BASELINE config 0's size)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkmi_loader import load_pkg  # noqa: E402


def main():
    out = sys.argv[1]
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    os.makedirs(out, exist_ok=True)
    zk = load_pkg().Zkmi()
    from test_cpu_host import _note_update_case

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
    print("wrote relation.bin (%d bytes) and witness.bin to %s" % (len(blob), out))


if __name__ == "__main__":
    main()

"""ORACLE — TEST INFRASTRUCTURE ONLY.

arkworks' CanonicalSerialize byte layout of Groth16 keys over BLS12-381, restated from memory of crates that
are not in the reference tree (oracle/README.md rows 8 and 10): ark-groth16 0.4.0 data_structures.rs
(field order of VerifyingKey / ProvingKey / Proof), ark-serialize 0.4.2 (Vec<T> = u64 little-endian length +
elements), ark-bls12-381 0.4.0 curves/util.rs (zcash-style big-endian point encodings, compressed or not).
PARITY UNPINNED: nothing in /root/reference serialises a key; this file and csrc/arkworks.hip are two
independent restatements of the same remembered layout and are tested against each other."""
from . import bls12_381 as ec


def g1(pt, compressed):
    if compressed:
        return ec.g1_compress(pt)
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def g2(pt, compressed):
    if compressed:
        return ec.g2_compress(pt)
    if pt is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = pt
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


def vec(items, enc, compressed):
    return len(items).to_bytes(8, "little") + b"".join(enc(x, compressed) for x in items)


def verifying_key(vk_wire, n_pub, compressed):
    """vk_wire: this repository's vk layout (alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | n_pub x gamma_abc_g1, affine LE)."""
    out = g1(ec.g1_from_bytes(vk_wire[:96]), compressed)
    for k in range(3):
        out += g2(ec.g2_from_bytes(vk_wire[96 + 192 * k : 288 + 192 * k]), compressed)
    abc = [ec.g1_from_bytes(vk_wire[672 + 96 * i : 768 + 96 * i]) for i in range(n_pub)]
    return out + vec(abc, g1, compressed)


def proving_key(vk_wire, n_pub, key, compressed):
    """key: dict of wire-format byte strings (beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query)."""
    pts1 = lambda b: [ec.g1_from_bytes(b[96 * i : 96 * i + 96]) for i in range(len(b) // 96)]
    pts2 = lambda b: [ec.g2_from_bytes(b[192 * i : 192 * i + 192]) for i in range(len(b) // 192)]
    out = verifying_key(vk_wire, n_pub, compressed)
    out += g1(ec.g1_from_bytes(key["beta_g1"]), compressed) + g1(ec.g1_from_bytes(key["delta_g1"]), compressed)
    out += vec(pts1(key["a_query"]), g1, compressed) + vec(pts1(key["b_g1_query"]), g1, compressed)
    out += vec(pts2(key["b_g2_query"]), g2, compressed)
    out += vec(pts1(key["h_query"]), g1, compressed) + vec(pts1(key["l_query"]), g1, compressed)
    return out

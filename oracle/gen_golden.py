"""ORACLE — TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/*.json from the
pure-Python oracle (run from the repo root: `python -m oracle.gen_golden`).

What pins what:
  * constants.json      public BLS12-381 known answers (generators, compressed
                        generator encodings, 2^32-th root of unity) re-derived here;
  * mock_boundary.json  the values the reference's OWN tests pin at the
                        mocked_zk / contract boundary (scalar.rs:36-54,
                        mocked_zk/src/tests.rs:27-35, contract/merkle.rs:115-132),
                        recomputed with hashlib;
  * poseidon.json       Poseidon-5 constants digest, permutation / hash / Merkle vectors for
                        BLS12-381 Fr and BN254 Fr, plus the published BN254 known answers the
                        generator is pinned by (circomlib), and the update_note relation's
                        public values for one seeded instance;
  * bn254.json          BN254 G1 MSM / Fr NTT / KZG-commit vectors (halo2curves constants);
  * ntt_small.json, msm_small.json, groth16_n128.json, pairing.json
                        oracle outputs on seeded inputs (parity unpinned against the
                        reference, which has no such code; pinned by the O(N^2)
                        DFT definition, naive double-and-add, and pairing
                        self-verification of the proof).
"""
import hashlib
import json
import os

from . import bls12_381 as ec
from . import groth16 as g
from . import ntt as nt
from .bls12_381 import R, Fq, Fq2

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def constants():
    dump(
        "constants.json",
        {
            "r": hex(R),
            "p": hex(ec.P),
            "fr_root_2_32": hex(ec.FR_ROOT_2_32),
            "g1": ec.g1_to_bytes(ec.G1).hex(),
            "g2": ec.g2_to_bytes(ec.G2).hex(),
            "g1_compressed": ec.g1_compress(ec.G1).hex(),
            "g2_compressed": ec.g2_compress(ec.G2).hex(),
            "g1_times_r_minus_1": ec.g1_to_bytes(ec.g1_mul(R - 1)).hex(),
            "g1_times_c0ffee": ec.g1_to_bytes(ec.g1_mul(0xC0FFEE)).hex(),
            "g2_times_c0ffee": ec.g2_to_bytes(ec.g2_mul(0xC0FFEE)).hex(),
            "synthetic_g1_first4": [ec.g1_to_bytes(p).hex() for p in ec.synthetic_bases_g1(4)],
            "synthetic_g1_index_1000": ec.g1_to_bytes(
                ec.pt_add(Fq, ec.G1, ec.g1_mul(1000 * 0xC0FFEE))
            ).hex(),
            "synthetic_g2_first4": [ec.g2_to_bytes(p).hex() for p in ec.synthetic_bases_g2(4)],
        },
    )


def mock_boundary():
    sha = lambda b: hashlib.sha256(b).digest()
    zero = bytes(32)
    acc_hash = sha(zero)
    note_hash = sha(zero + zero + zero + acc_hash)
    s1 = (1).to_bytes(16, "little") + bytes(16)
    s2 = (2).to_bytes(16, "little") + bytes(16)
    cur = sha(s1 + s2)
    for _ in range(9):
        cur = sha(cur + zero)
    dump(
        "mock_boundary.json",
        {
            "scalar_from_123456_prefix": "40e201",  # scalar.rs:36-43
            "scalar_to_u128_bytes": "b168de3a",  # scalar.rs:45-54 -> 987654321
            "scalar_to_u128_value": 987654321,
            "account_hash_empty": acc_hash.hex(),  # account.rs:16-24
            "empty_note_hash": note_hash.hex(),  # mocked_zk/src/tests.rs:27-35
            "merkle_root_two_leaves": cur.hex(),  # contract/merkle.rs:115-132
        },
    )


def ntt_small():
    cases = []
    rng = ec.SplitMix64(0x5A4B0001)
    for log_n in (1, 2, 3, 5, 8, 10):
        n = 1 << log_n
        a = [rng.fr() for _ in range(n)]
        if log_n == 3:
            a[0], a[1], a[2] = 0, 1, R - 1
        fwd = nt.ntt(a)
        if log_n <= 8:
            assert fwd == nt.dft_naive(a)
        cases.append(
            {
                "log_n": log_n,
                "input": b"".join(map(ec.fr_to_bytes, a)).hex(),
                "forward": b"".join(map(ec.fr_to_bytes, fwd)).hex(),
                "inverse": b"".join(map(ec.fr_to_bytes, nt.ntt(a, inverse=True))).hex(),
                "coset_forward": b"".join(map(ec.fr_to_bytes, nt.coset_ntt(a))).hex(),
                "coset_inverse": b"".join(map(ec.fr_to_bytes, nt.coset_intt(a))).hex(),
            }
        )
    dump("ntt_small.json", cases)


def msm_small():
    rng = ec.SplitMix64(0x5A4B0002)
    cases = []
    bases1 = ec.synthetic_bases_g1(64)
    bases2 = ec.synthetic_bases_g2(24)

    def case(name, group, scalars, bases):
        F = Fq if group == 1 else Fq2
        tob = ec.g1_to_bytes if group == 1 else ec.g2_to_bytes
        res = ec.msm_naive(F, scalars, bases)
        cases.append(
            {
                "name": name,
                "group": group,
                "scalars": b"".join(map(ec.fr_to_bytes, scalars)).hex(),
                "bases": b"".join(map(tob, bases)).hex(),
                "expected": tob(res).hex(),
            }
        )

    case("g1_single_one", 1, [1], bases1[:1])
    case("g1_single_r_minus_1", 1, [R - 1], bases1[:1])
    case("g1_all_zero_scalars", 1, [0] * 8, bases1[:8])
    case("g1_uniform_64", 1, [rng.fr() for _ in range(64)], bases1)
    case("g1_small_scalars", 1, [i for i in range(33)], bases1[:33])
    case("g1_repeated_base", 1, [rng.fr() for _ in range(16)], [bases1[3]] * 16)
    case("g1_cancelling", 1, [5, R - 5, 7], [bases1[1], bases1[1], bases1[2]])
    case("g1_with_infinity_bases", 1, [rng.fr() for _ in range(6)], [bases1[0], None, bases1[2], None, None, bases1[5]])
    # witness-like: 40% zero, 20% one, 10% < 2^16, 30% uniform
    wl = []
    for _ in range(64):
        t = rng.next() % 10
        wl.append(0 if t < 4 else 1 if t < 6 else (rng.next() & 0xFFFF) if t < 7 else rng.fr())
    case("g1_witness_like_64", 1, wl, bases1)
    # window-boundary stress: all-ones digits, 2^k and 2^k - 1
    case("g1_digit_edges", 1, [(1 << 255) % R, (1 << 254) - 1, (1 << 16) - 1, 1 << 15, (1 << 15) + 1, R - 2], bases1[:6])
    case("g2_single_one", 2, [1], bases2[:1])
    case("g2_uniform_24", 2, [rng.fr() for _ in range(24)], bases2)
    case("g2_with_infinity_and_zero", 2, [0, rng.fr(), rng.fr(), R - 1], [bases2[0], None, bases2[2], bases2[3]])
    dump("msm_small.json", cases)


def pairing_kat():
    a, b = 0x1234567, 0x7654321
    e = ec.pairing(ec.g1_mul(a), ec.g2_mul(b))
    assert e == ec.f12_pow(ec.pairing(ec.G1, ec.G2), a * b)
    dump(
        "pairing.json",
        {
            "basis": "Fq[w]/(w^12 - 2w^6 + 2), coefficient of w^i at index i",
            "a": a,
            "b": b,
            "e_g1_g2": [hex(v) for v in ec.pairing(ec.G1, ec.G2)],
            "e_aG1_bG2": [hex(v) for v in e],
        },
    )


def groth16_n128():
    log_n = 7
    r1 = g.shielder_r1cs(log_n)
    seed = 0x5A4B0000
    z = g.shielder_witness(log_n, seed)
    assert r1.is_satisfied(z)
    rng = ec.SplitMix64(0x5A4B00AA)
    tau, alpha, beta, gamma, delta, r, s = [rng.fr() for _ in range(7)]
    pk, vk = g.setup(r1, tau, alpha, beta, gamma, delta)
    h = g.witness_map(r1, z)
    proof = g.prove(pk, r1, z, r, s)
    assert g.verify(vk, z[1 : r1.n_pub], proof)
    vk_bytes = (
        ec.g1_to_bytes(vk["alpha_g1"])
        + ec.g2_to_bytes(vk["beta_g2"])
        + ec.g2_to_bytes(vk["gamma_g2"])
        + ec.g2_to_bytes(vk["delta_g2"])
        + b"".join(map(ec.g1_to_bytes, vk["gamma_abc_g1"]))
    )
    dump(
        "groth16_n128.json",
        {
            "log_n": log_n,
            "witness_seed": seed,
            "toxic": b"".join(map(ec.fr_to_bytes, (tau, alpha, beta, gamma, delta))).hex(),
            "r": ec.fr_to_bytes(r).hex(),
            "s": ec.fr_to_bytes(s).hex(),
            "witness": b"".join(map(ec.fr_to_bytes, z)).hex(),
            "h": b"".join(map(ec.fr_to_bytes, h)).hex(),
            "vk": vk_bytes.hex(),
            "pk": {
                "alpha_g1": ec.g1_to_bytes(pk["alpha_g1"]).hex(),
                "beta_g1": ec.g1_to_bytes(pk["beta_g1"]).hex(),
                "beta_g2": ec.g2_to_bytes(pk["beta_g2"]).hex(),
                "delta_g1": ec.g1_to_bytes(pk["delta_g1"]).hex(),
                "delta_g2": ec.g2_to_bytes(pk["delta_g2"]).hex(),
                "a_query": b"".join(map(ec.g1_to_bytes, pk["a_query"])).hex(),
                "b_g1_query": b"".join(map(ec.g1_to_bytes, pk["b_g1_query"])).hex(),
                "b_g2_query": b"".join(map(ec.g2_to_bytes, pk["b_g2_query"])).hex(),
                "h_query": b"".join(map(ec.g1_to_bytes, pk["h_query"])).hex(),
                "l_query": b"".join(map(ec.g1_to_bytes, pk["l_query"])).hex(),
            },
            "proof": g.proof_to_bytes(proof).hex(),
        },
    )


def poseidon_vectors():
    from . import poseidon as ps

    out = {"published_bn254": {
        "rc0_t2_8_56": hex(ps.spec("bn254_fr", 2, 8, 56)[0][0][0]),
        "rc0_t3_8_57": hex(ps.spec("bn254_fr", 3, 8, 57)[0][0][0]),
        "permute_0_1_2_t3_8_57": hex(ps.permute([0, 1, 2], "bn254_fr", 8, 57)[0]),
        "permute_0_1_2_3_4_t5_8_60": hex(ps.permute([0, 1, 2, 3, 4], "bn254_fr", 8, 60)[0]),
    }}
    for field in ("bls12_381_fr", "bn254_fr"):
        p, _ = ps.FIELDS[field]
        rc, mds = ps.spec(field)
        flat = b"".join(v.to_bytes(32, "little") for row in rc for v in row) + b"".join(
            v.to_bytes(32, "little") for row in mds for v in row)
        rng = ec.SplitMix64(0x905E1D00 + len(field))
        cases = []
        for arity in (0, 1, 2, 3, 4, 5, 8):
            vals = [rng.next() * rng.next() * rng.next() * rng.next() % p for _ in range(arity)]
            cases.append({"inputs": [hex(v) for v in vals], "hash": hex(ps.hash_fix_len(vals, field))})
        leaves = [ps.hash_fix_len([i], field) for i in range(8)]
        out[field] = {
            "constants_sha256": hashlib.sha256(flat).hexdigest(),
            "rc_first": hex(rc[0][0]), "rc_last": hex(rc[-1][-1]), "mds_00": hex(mds[0][0]), "mds_44": hex(mds[4][4]),
            "permute_0_1_2_3_4": [hex(v) for v in ps.permute([0, 1, 2, 3, 4], field)],
            "hashes": cases,
            "merkle_8_leaves_root": hex(ps.merkle_tree(leaves, field)[-1][0]),
        }
    # one update_note instance (withdraw): inputs and the public values the relation exposes
    rng = ec.SplitMix64(0x0DA7E)
    tok = [rng.fr(), rng.fr()]
    bal = [1000, 77]
    new_id, old_id, ot, on, nt_, nn, user = (rng.fr() for _ in range(7))
    shape = [rng.next() & 1 for _ in range(10)]
    path = [rng.fr() for _ in range(10)]
    amount = 250
    old_acc = ps.hash_fix_len([tok[0], bal[0], tok[1], bal[1]])
    root = ps.merkle_root(ps.hash_fix_len([old_id, ot, on, old_acc]), shape, path)
    new_acc = ps.hash_fix_len([tok[0], bal[0] - amount, tok[1], bal[1]])
    out["update_note_withdraw"] = {
        "amount": amount, "token": hex(tok[0]), "user": hex(user),
        "new_note": [hex(new_id), hex(nt_), hex(nn)], "old_note": [hex(old_id), hex(ot), hex(on)],
        "path_shape": shape, "path": [hex(v) for v in path],
        "account": [hex(tok[0]), hex(bal[0]), hex(tok[1]), hex(bal[1])],
        "publics": [hex(v) for v in (amount, tok[0], user, ps.hash_fix_len([new_id, nt_, nn, new_acc]), root, on)],
    }
    dump("poseidon.json", out)


def bn254_vectors():
    from . import bn254 as bn

    rng = ec.SplitMix64(0xB254)
    rnd = lambda: rng.next() * rng.next() * rng.next() * rng.next() % bn.R
    n = 64
    pts = bn.synthetic_bases(n)
    sc = [rnd() for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, bn.R - 1
    a = [rnd() for _ in range(32)]
    frs = lambda v: b"".join(int(x).to_bytes(32, "little") for x in v).hex()
    tau = 0xDEADBEEFCAFEF00D1234567 % bn.R
    coeffs = [rnd() for _ in range(16)]
    zeta = rnd()  # (drawn after everything the earlier vectors use)
    gp_num, gp_den = [rnd() for _ in range(12)], [rnd() or 1 for _ in range(12)]
    srs_pts = [bn.pt_mul(bn.G1, pow(tau, i, bn.R)) for i in range(16)]
    open_eval, open_proof = bn.kzg_open(coeffs, zeta, srs_pts)
    gp_z, gp_total = bn.grand_product(gp_num, gp_den)
    dump("bn254.json", {
        "p": hex(bn.P), "r": hex(bn.R), "root_2_28": hex(bn.FR_ROOT_2_28),
        "synthetic_first4": [bn.g1_to_bytes(p).hex() for p in pts[:4]],
        "msm": {"scalars": frs(sc), "bases": b"".join(bn.g1_to_bytes(p) for p in pts).hex(),
                "expected": bn.g1_to_bytes(bn.msm_naive(sc, pts)).hex()},
        "ntt": {"log_n": 5, "input": frs(a), "forward": frs(bn.ntt(a)), "inverse": frs(bn.ntt(a, inverse=True)),
                "coset_forward": frs(bn.ntt(a, coset=True)), "coset_inverse": frs(bn.ntt(a, inverse=True, coset=True))},
        "kzg": {"log_n": 4, "tau": hex(tau),
                "srs": b"".join(bn.g1_to_bytes(bn.pt_mul(bn.G1, pow(tau, i, bn.R))) for i in range(16)).hex(),
                "evaluations": frs(bn.ntt(coeffs)), "coefficients": frs(coeffs),
                "commitment": bn.g1_to_bytes(bn.pt_mul(bn.G1, sum(c * pow(tau, i, bn.R) for i, c in enumerate(coeffs)) % bn.R)).hex()},
        # opening of the same polynomial against the same SRS (eval_polynomial, kate_division, commit(q))
        "kzg_open": {"zeta": frs([zeta]), "eval": frs([open_eval]), "quotient": frs(bn.kate_division(coeffs, zeta)),
                     "proof": bn.g1_to_bytes(open_proof).hex()},
        "grand_product": {"num": frs(gp_num), "den": frs(gp_den), "z": frs(gp_z), "total": frs([gp_total])},
    })


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    poseidon_vectors()
    bn254_vectors()
    constants()
    mock_boundary()
    ntt_small()
    msm_small()
    pairing_kat()
    groth16_n128()

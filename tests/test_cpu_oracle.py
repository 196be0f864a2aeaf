"""CPU suite, part 1: the oracle against the golden vectors and its own
definitions (no GPU, no product code)."""
import hashlib

from conftest import golden
from oracle import bls12_381 as ec
from oracle import cpp as ocpp
from oracle import groth16 as g16
from oracle import ntt as ont
from oracle.bls12_381 import P, R

H = bytes.fromhex


def test_curve_constants_known_answers():
    c = golden("constants.json")
    x = -0xD201000000010000
    assert int(c["r"], 16) == x**4 - x**2 + 1 == R
    assert int(c["p"], 16) == (x - 1) ** 2 * R // 3 + x == P
    assert ec.on_curve(ec.Fq, ec.B_G1, ec.G1) and ec.on_curve(ec.Fq2, ec.B_G2, ec.G2)
    assert ec.pt_mul(ec.Fq, ec.G1, R) is None and ec.pt_mul(ec.Fq2, ec.G2, R) is None
    # public KAT: zcash-format compressed generators
    assert c["g1_compressed"] == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
        "6c55e83ff97a1aeffb3af00adb22c6bb"
    )
    # the full 96-byte compressed G2 generator (zcash bls12_381 crate / IETF pairing-friendly-curves draft, appendix)
    assert c["g2_compressed"] == (
        "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    )
    assert ec.g1_compress(ec.G1).hex() == c["g1_compressed"]
    assert ec.g2_compress(ec.G2).hex() == c["g2_compressed"]
    w = int(c["fr_root_2_32"], 16)
    assert w == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    assert pow(w, 1 << 32, R) == 1 and pow(w, 1 << 31, R) == R - 1


PUBLIC_KATS = {
    # eth2 interop / BLS-signature public keys of the secret keys 1, 2, 3 = compressed [k]G1 (widely published)
    "g1_x1": "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
    "g1_x2": "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e",
    "g1_x3": "89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224",
    # -G1: same x, sign bit set (zcash encoding: bit 5 of the first byte = y is the lexicographically larger root)
    "g1_neg": "b7f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
    # [2]G2, second entry of the zcash crate's g2_compressed_valid_test_vectors
    "g2_x2": "aa4edef9c1ed7f729f520e47730a124fd70662a904ba1074728114d1031e1572c6c886f6b57ec72a6178288c47c33577"
             "1638533957d540a9d2370f17cc7ed5863bc0b995b8825e0ee1ea1e1e4d00dbae81f14b0bf3611b78c952aacab827a053",
}
# G2 generator coordinates (IETF draft-irtf-cfrg-pairing-friendly-curves, BLS12-381): x = x0 + x1 u, y = y0 + y1 u
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)


def test_public_point_encoding_known_answers():
    """Published compressed encodings of small multiples of the generators, recalled from public sources that need no
    toolchain (each labelled above): they pin scalar multiplication, the sign convention and the Fq2 component order
    (c1 first) of the zcash / ark-bls12-381 encoding in the oracle.  tests/test_cpu_host.py repeats them on the product's
    host code."""
    assert ec.g1_compress(ec.G1).hex() == PUBLIC_KATS["g1_x1"]
    assert ec.g1_compress(ec.g1_mul(2)).hex() == PUBLIC_KATS["g1_x2"]
    assert ec.g1_compress(ec.g1_mul(3)).hex() == PUBLIC_KATS["g1_x3"]
    assert ec.g1_compress(ec.g1_mul(R - 1)).hex() == PUBLIC_KATS["g1_neg"]
    assert ec.g2_compress(ec.g2_mul(2)).hex() == PUBLIC_KATS["g2_x2"]
    assert (tuple(ec.G2[0]), tuple(ec.G2[1])) == G2_GEN
    assert ec.g1_compress(None).hex() == "c0" + "00" * 47 and ec.g2_compress(None).hex() == "c0" + "00" * 95


def test_reference_pinned_mock_boundary_vectors():
    """Values the reference's own tests pin."""
    m = golden("mock_boundary.json")
    assert (123456).to_bytes(16, "little")[:3].hex() == m["scalar_from_123456_prefix"]
    assert int.from_bytes(H(m["scalar_to_u128_bytes"]) + bytes(12), "little") == m["scalar_to_u128_value"]
    assert hashlib.sha256(bytes(32)).hexdigest() == m["account_hash_empty"]
    assert m["account_hash_empty"] == "66687aadf862bd776c8fc18b8e9f8e20089714856ee233b3902a591d0d5f2925"
    assert m["empty_note_hash"] == "09ed9cfa36a525a93385ccdb9c324cd5e7c6396f2bc1f00860a3a7ad97e827c6"
    assert m["merkle_root_two_leaves"] == "1025a722a6336773a341a72fdb484a51b003b588638729623e8ff89fbbfffd59"


def test_py_ntt_matches_definition():
    rng = ec.SplitMix64(9)
    for lg in (1, 3, 6):
        a = [rng.fr() for _ in range(1 << lg)]
        assert ont.ntt(a) == ont.dft_naive(a)
        assert ont.ntt(a, inverse=True) == ont.dft_naive(a, inverse=True)
        assert ont.coset_intt(ont.coset_ntt(a)) == a
    # convolution theorem
    a = [rng.fr() for _ in range(8)] + [0] * 8
    b = [rng.fr() for _ in range(8)] + [0] * 8
    c = ont.ntt([x * y % R for x, y in zip(ont.ntt(a), ont.ntt(b))], inverse=True)
    exp = [0] * 16
    for i in range(8):
        for j in range(8):
            exp[i + j] = (exp[i + j] + a[i] * b[j]) % R
    assert c == exp


def test_cpp_oracle_ntt_golden():
    for c in golden("ntt_small.json"):
        x, lg = H(c["input"]), c["log_n"]
        assert ocpp.ntt(x, lg) == H(c["forward"])
        assert ocpp.ntt(x, lg, True) == H(c["inverse"])
        assert ocpp.ntt(x, lg, False, True) == H(c["coset_forward"])
        assert ocpp.ntt(x, lg, True, True) == H(c["coset_inverse"])


def test_cpp_oracle_msm_golden():
    for c in golden("msm_small.json"):
        f = ocpp.msm_g1 if c["group"] == 1 else ocpp.msm_g2
        assert f(H(c["scalars"]), H(c["bases"])) == H(c["expected"]), c["name"]


def test_cpp_oracle_thread_count_invariance():
    c = [x for x in golden("msm_small.json") if x["name"] == "g1_uniform_64"][0]
    assert ocpp.msm_g1(H(c["scalars"]), H(c["bases"]), 1) == ocpp.msm_g1(H(c["scalars"]), H(c["bases"]), 3)
    n = golden("ntt_small.json")[-1]
    assert ocpp.ntt(H(n["input"]), n["log_n"], nthreads=1) == ocpp.ntt(H(n["input"]), n["log_n"], nthreads=5)


def test_py_pairing_bilinear_and_golden():
    pg = golden("pairing.json")
    e = ec.pairing(ec.G1, ec.G2)
    assert [hex(v) for v in e] == pg["e_g1_g2"]
    assert e != ec.f12_one() and ec.f12_pow(e, R) == ec.f12_one()


def test_py_groth16_golden_proof_verifies():
    gd = golden("groth16_n128.json")
    r1 = g16.shielder_r1cs(gd["log_n"])
    z = g16.shielder_witness(gd["log_n"], gd["witness_seed"])
    assert b"".join(map(ec.fr_to_bytes, z)) == H(gd["witness"])
    assert r1.is_satisfied(z)
    vk = H(gd["vk"])
    vkd = {
        "alpha_g1": ec.g1_from_bytes(vk[:96]),
        "beta_g2": ec.g2_from_bytes(vk[96:288]),
        "gamma_g2": ec.g2_from_bytes(vk[288:480]),
        "delta_g2": ec.g2_from_bytes(vk[480:672]),
        "gamma_abc_g1": [ec.g1_from_bytes(vk[672 + 96 * i : 768 + 96 * i]) for i in range(r1.n_pub)],
    }
    proof = g16.proof_from_bytes(H(gd["proof"]))
    assert g16.verify(vkd, z[1 : r1.n_pub], proof)
    bad = list(z[1 : r1.n_pub])
    bad[3] = (bad[3] + 1) % R
    assert not g16.verify(vkd, bad, proof)


def test_cpp_oracle_prover_golden(zk):
    """C++ restatement reproduces the Python oracle's proof bytes (uses the
    product only to export the relation's CSR matrices)."""
    gd = golden("groth16_n128.json")
    r1 = zk.shielder_r1cs(gd["log_n"])
    mats = [r1.export(m) for m in range(3)]
    pk = {k: H(v) for k, v in gd["pk"].items()}
    wit = H(gd["witness"])
    assert ocpp.witness_map(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, wit) == H(gd["h"])
    pf = ocpp.groth16_prove(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, pk, wit, H(gd["r"]), H(gd["s"]))
    assert pf == H(gd["proof"])


def test_cpp_oracle_setup_golden(zk):
    """The C++ restatement of the trusted setup (arkworks generator shape) reproduces the Python oracle's
    verifying key and all five queries at N = 128, so it can stand in for it at sizes Python cannot reach
    (the GPU suite checks the product's setup against it at 2^16)."""
    gd = golden("groth16_n128.json")
    r1 = zk.shielder_r1cs(gd["log_n"])
    mats = [r1.export(m) for m in range(3)]
    vk, key = ocpp.groth16_setup(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, H(gd["toxic"]))
    assert vk == H(gd["vk"])
    for name, want in gd["pk"].items():
        assert key[name] == H(want), name
    # thread count does not change the result
    vk1, key1 = ocpp.groth16_setup(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, H(gd["toxic"]), nthreads=1)
    assert vk1 == vk and key1 == key


# ---- Poseidon ------------------------------------------------------------


def test_poseidon_generator_matches_published_bn254_vectors():
    """The Grain constant generator, the Cauchy MDS and the permutation reproduce circomlib's
    published BN254 constants / test vectors (quoted from memory of circomlib's
    poseidon_constants and test/poseidoncircuit.js; the same procedure pse-poseidon runs)."""
    from oracle import poseidon as ps

    assert ps.spec("bn254_fr", 2, 8, 56)[0][0][0] == 0x09C46E9EC68E9BD4FE1FAABA294CBA38A71AA177534CDD1B6C7DC0DBD0ABD7A7
    assert ps.spec("bn254_fr", 3, 8, 57)[0][0][0] == 0x0EE9A592BA9A9518D05986D656F40C2114C4993C11BB29938D21D47304CD8E6E
    assert ps.permute([0, 1, 2], "bn254_fr", 8, 57)[0] == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A
    assert ps.permute([0, 1, 2, 3, 4], "bn254_fr", 8, 60)[0] == 0x299C867DB6C1FDD79DCEFA40E4510B9837E60EBB1CE0663DBAA525DF65250465


def test_poseidon_spec_shape_and_mds_invertible():
    from oracle import poseidon as ps

    for field in ("bls12_381_fr", "bn254_fr"):
        p, _ = ps.FIELDS[field]
        rc, mds = ps.spec(field)
        assert len(rc) == ps.R_F + ps.R_P and all(len(r) == ps.T_WIDTH for r in rc)
        assert all(0 <= c < p for r in rc for c in r)
        # Cauchy matrices are invertible: Gaussian elimination finds 5 pivots
        m = [list(r) for r in mds]
        for col in range(5):
            piv = next(i for i in range(col, 5) if m[i][col] % p)
            m[col], m[piv] = m[piv], m[col]
            inv = pow(m[col][col], -1, p)
            for i in range(col + 1, 5):
                f = m[i][col] * inv % p
                m[i] = [(a - f * b) % p for a, b in zip(m[i], m[col])]


def test_poseidon_sponge_framing_and_merkle_path():
    from oracle import poseidon as ps

    p = ps.BLS_FR
    # 4 inputs = RATE: one absorbing permutation + one padding-only permutation
    st = ps.permute([(1 << 64), 1, 2, 3, 4])
    st[1] = (st[1] + 1) % p
    assert ps.hash_fix_len([1, 2, 3, 4]) == ps.permute(st)[1]
    # 2 inputs: a single permutation with the padding 1 behind the inputs
    assert ps.hash_fix_len([7, 9]) == ps.permute([(1 << 64), 7, 9, 1, 0])[1]
    # Merkle path of leaf 5 in a 16-leaf tree recomputes the root (merkle_proof.rs:38-61)
    leaves = [ps.hash_fix_len([i, i + 1]) for i in range(16)]
    levels = ps.merkle_tree(leaves)
    idx = 5
    shape = [1 - ((idx >> lv) & 1) for lv in range(4)]
    path = [levels[lv][(idx >> lv) ^ 1] for lv in range(4)]
    assert ps.merkle_root(leaves[idx], shape, path) == levels[-1][0]


# ---- BN254 ------------------------------------------------------------------


def test_bn254_constants_and_definitions():
    from oracle import bn254 as bn

    assert bn.on_curve(bn.G1) and bn.pt_mul(bn.G1, bn.R) is None
    assert pow(2, bn.P - 1, bn.P) == 1 and pow(2, bn.R - 1, bn.R) == 1
    # halo2curves bn256::Fr::ROOT_OF_UNITY = 7^((r-1)/2^28), primitive of order 2^28
    assert pow(bn.FR_GENERATOR, (bn.R - 1) >> 28, bn.R) == bn.FR_ROOT_2_28
    assert pow(bn.FR_ROOT_2_28, 1 << 27, bn.R) == bn.R - 1
    rng = ec.SplitMix64(3)
    a = [rng.next() * rng.next() % bn.R for _ in range(32)]
    assert bn.ntt(a) == bn.dft_naive(a)
    assert bn.ntt(a, inverse=True) == bn.dft_naive(a, inverse=True)
    assert bn.ntt(bn.ntt(a, coset=True), inverse=True, coset=True) == a
    pts = bn.synthetic_bases(5)
    assert pts[3] == bn.pt_mul(bn.G1, 1 + 3 * 0xC0FFEE) and all(bn.on_curve(p) for p in pts)


def test_poseidon_and_bn254_golden_fixtures_reproduce():
    """tests/golden/poseidon.json and bn254.json (python -m oracle.gen_golden) are what the oracle
    computes today; the published BN254 Poseidon values in the fixture are the literal known answers."""
    from oracle import bn254 as bn
    from oracle import poseidon as ps

    g = golden("poseidon.json")
    pub = g["published_bn254"]
    assert pub["rc0_t2_8_56"] == "0x9c46e9ec68e9bd4fe1faaba294cba38a71aa177534cdd1b6c7dc0dbd0abd7a7"
    assert pub["permute_0_1_2_3_4_t5_8_60"] == "0x299c867db6c1fdd79dcefa40e4510b9837e60ebb1ce0663dbaa525df65250465"
    for field in ("bls12_381_fr", "bn254_fr"):
        f = g[field]
        assert [hex(v) for v in ps.permute([0, 1, 2, 3, 4], field)] == f["permute_0_1_2_3_4"]
        for c in f["hashes"]:
            assert hex(ps.hash_fix_len([int(v, 16) for v in c["inputs"]], field)) == c["hash"]
    b = golden("bn254.json")
    sc = [int.from_bytes(H(b["msm"]["scalars"])[i : i + 32], "little") for i in range(0, 64 * 32, 32)]
    assert bn.g1_to_bytes(bn.msm_naive(sc, bn.synthetic_bases(64))).hex() == b["msm"]["expected"]
    a = [int.from_bytes(H(b["ntt"]["input"])[i : i + 32], "little") for i in range(0, 32 * 32, 32)]
    assert b"".join(v.to_bytes(32, "little") for v in bn.ntt(a)).hex() == b["ntt"]["forward"]
    ints = lambda h: [int.from_bytes(H(h)[i : i + 32], "little") for i in range(0, len(h) // 2, 32)]
    frs = lambda v: b"".join(x.to_bytes(32, "little") for x in v).hex()
    co, o = ints(b["kzg"]["coefficients"]), b["kzg_open"]
    z = ints(o["zeta"])[0]
    assert (frs([bn.eval_polynomial(co, z)]), frs(bn.kate_division(co, z))) == (o["eval"], o["quotient"])
    tau = int(b["kzg"]["tau"], 16)  # the proof is [q(tau)] G
    assert bn.g1_to_bytes(bn.pt_mul(bn.G1, bn.eval_polynomial(ints(o["quotient"]), tau))).hex() == o["proof"]
    gp = b["grand_product"]
    zs, tot = bn.grand_product(ints(gp["num"]), ints(gp["den"]))
    assert (frs(zs), frs([tot])) == (gp["z"], gp["total"])


def test_bn254_kzg_opening_oracle_identities():
    """eval_polynomial / kate_division / kzg_open restate halo2's arithmetic: p(X) = (X - z) q(X) + p(z) coefficient by
    coefficient, and against an SRS with a KNOWN tau the opening is [q(tau)] G with (tau - z) q(tau) = p(tau) - p(z) -- the
    relation a verifier checks through the pairing."""
    import random

    from oracle import bn254 as bn

    rnd = random.Random(254)
    for n in (1, 2, 3, 17, 64):
        p = [rnd.randrange(bn.R) for _ in range(n)]
        z = rnd.randrange(bn.R)
        y, q = bn.eval_polynomial(p, z), bn.kate_division(p, z)
        assert len(q) == n - 1
        assert y == sum(c * pow(z, i, bn.R) for i, c in enumerate(p)) % bn.R
        back = [0] * n  # (X - z) q + y
        for i, c in enumerate(q):
            back[i + 1] = (back[i + 1] + c) % bn.R
            back[i] = (back[i] - z * c) % bn.R
        back[0] = (back[0] + y) % bn.R
        assert back == p
    tau = 0x1234567DEADBEEF % bn.R
    n = 12
    srs = [bn.pt_mul(bn.G1, pow(tau, i, bn.R)) for i in range(n)]
    p = [rnd.randrange(bn.R) for _ in range(n)]
    z = rnd.randrange(bn.R)
    y, proof = bn.kzg_open(p, z, srs)
    q_tau = bn.eval_polynomial(bn.kate_division(p, z), tau)
    assert proof == bn.pt_mul(bn.G1, q_tau)
    assert (tau - z) * q_tau % bn.R == (bn.eval_polynomial(p, tau) - y) % bn.R
    # several polynomials at one point: one proof for f = sum_j v^j p_j, checked against the folded evaluations
    polys, v = [[rnd.randrange(bn.R) for _ in range(n)] for _ in range(4)], rnd.randrange(bn.R)
    ys, proof = bn.kzg_open_many(polys, z, v, srs)
    assert ys == [bn.eval_polynomial(p, z) for p in polys]
    f = [sum(pow(v, j, bn.R) * p[i] for j, p in enumerate(polys)) % bn.R for i in range(n)]
    q_tau = bn.eval_polynomial(bn.kate_division(f, z), tau)
    assert proof == bn.pt_mul(bn.G1, q_tau)
    assert (tau - z) * q_tau % bn.R == sum(pow(v, j, bn.R) * (bn.eval_polynomial(p, tau) - y) for j, (p, y) in enumerate(zip(polys, ys))) % bn.R



def test_quad_split_addition_dataflow_model():
    """The round table of csrc/quad.hpp (one XYZZ coordinate per lane of a quad, four rounds of one product
    per lane, operands exchanged by quad_perm selectors) as a plain-integer model: four "lanes" holding numbers mod p, the same
    selector constants, the same per-lane operand choices -- against add-2008-s / dbl-2008-s-1 written out directly, and against
    the affine group law on real curve points.  It pins the DATAFLOW (which lane multiplies what, which selector moves which
    value); the HIP code itself is checked on the GPU by test_quad_split_addition_selftest."""
    import random

    from oracle import bls12_381 as ec

    p = ec.P
    perm = lambda sel, v: [v[sel[i]] for i in range(4)]
    ROT2, SWAP, B0, B1 = [2, 3, 0, 1], [1, 0, 3, 2], [0] * 4, [1] * 4
    P0023, P0110, P0223, P3103 = [0, 0, 2, 3], [0, 1, 1, 0], [0, 2, 2, 3], [3, 1, 0, 3]
    mul = lambda a, b: [x * y % p for x, y in zip(a, b)]
    sub = lambda a, b: [(x - y) % p for x, y in zip(a, b)]
    K = range(4)

    def quad_dbl(v):
        a1 = [v[k] * 2 % p if k == 1 else v[k] for k in K]
        m1 = mul(a1, a1)
        m = [3 * x % p for x in perm(B0, m1)]
        vv = perm(B1, m1)
        m2 = mul([m[k] if k == 3 else a1[k] for k in K], [m[k] if k == 3 else vv[k] for k in K])
        x3 = [(perm(P3103, m2)[k] - 2 * m2[k]) % p for k in K]
        w = perm(B1, m2)
        m3 = mul([m[k] if k == 0 else v[k] for k in K], [(m2[k] - x3[k]) % p if k == 0 else w[k] for k in K])
        y3 = sub(perm(P0023, m3), m3)
        return [x3[0], y3[1], m2[2], m3[3]]

    def quad_add(v, o):
        if o[2] == 0:
            return list(v)
        if v[2] == 0:
            return list(o)
        m1 = mul(v, perm(ROT2, o))
        d = sub(perm(ROT2, m1), m1)
        m2 = mul([d[k] if k < 2 else v[k] for k in K], [d[k] if k < 2 else o[k] for k in K])
        if m2[0] == 0:
            return quad_dbl(v) if m2[1] == 0 else [0, 0, 0, 0]
        pp, u1 = perm(B0, m2), perm(P0023, m1)
        m3 = mul([d[0], u1[1], m2[2], m2[3]], pp)
        s2, s3 = perm(SWAP, m2), perm(SWAP, m3)
        x3 = [((s2[k] if k == 0 else m2[k]) - (m3[k] if k == 0 else s3[k]) - 2 * (s3[k] if k == 0 else m3[k])) % p for k in K]
        s1p = perm(P0110, [d[k] if k == 0 else m1[k] for k in K])
        ppp = perm(B0, m3)
        m4 = mul([m3[0], d[1], s1p[2], m3[3]], [s1p[0], (m3[1] - x3[1]) % p, ppp[2], s1p[3]])
        y3 = sub(m4, perm(P0223, m4))
        return [x3[0], y3[1], m3[2], m4[3]]

    def ref_add(a, o):
        x1, y1, zz1, zzz1 = a
        x2, y2, zz2, zzz2 = o
        u1, u2, s1, s2 = x1 * zz2 % p, x2 * zz1 % p, y1 * zzz2 % p, y2 * zzz1 % p
        P_, R_ = (u2 - u1) % p, (s2 - s1) % p
        PP = P_ * P_ % p
        PPP, Q = P_ * PP % p, u1 * PP % p
        x3 = (R_ * R_ - PPP - 2 * Q) % p
        return [x3, (R_ * (Q - x3) - s1 * PPP) % p, zz1 * zz2 * PP % p, zzz1 * zzz2 * PPP % p]

    def ref_dbl(a):
        x, y, zz, zzz = a
        u = 2 * y % p
        v = u * u % p
        w, s, m = u * v % p, x * v % p, 3 * x * x % p
        x3 = (m * m - 2 * s) % p
        return [x3, (m * (s - x3) - w * y) % p, v * zz % p, w * zzz % p]

    rnd = random.Random(60)
    for _ in range(50):
        a, o = [rnd.randrange(p) for _ in range(4)], [rnd.randrange(p) for _ in range(4)]
        assert quad_add(a, o) == ref_add(a, o)
        assert quad_dbl(a) == ref_dbl(a)
        lam = rnd.randrange(1, p)
        same = [a[0] * lam**2 % p, a[1] * lam**3 % p, a[2] * lam**2 % p, a[3] * lam**3 % p]
        assert quad_add(a, same) == ref_dbl(a)  # equal points in different representations: the doubling rounds
        assert quad_add(a, [same[0], -same[1] % p, same[2], same[3]]) == [0, 0, 0, 0]
    # on the curve: k G + m G through the quad dataflow equals the affine law
    to_xyzz = lambda pt: [pt[0], pt[1], 1, 1]

    def to_affine(q):
        zi = pow(q[3], -1, p)
        zzi = pow(q[2], -1, p)
        return (q[0] * zzi % p, q[1] * zi % p)

    for k, m in ((1, 1), (2, 3), (5, 11), (123456789, 987654321)):
        pk, pm = ec.g1_mul(k), ec.g1_mul(m)
        got = quad_add(to_xyzz(pk), to_xyzz(pm))
        assert to_affine(got) == tuple(ec.g1_mul(k + m))
    acc = [0, 0, 0, 0]
    for k in range(1, 9):  # infinity + G + 2G + ... (the first addition takes the "this is infinity" exit)
        acc = quad_add(acc, to_xyzz(ec.g1_mul(k)))
    assert to_affine(acc) == tuple(ec.g1_mul(36))

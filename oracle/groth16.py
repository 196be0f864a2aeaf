"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Groth16 over BLS12-381 restated with Python integers (SURVEY.md §8a rows
a7, a10, a11), plus the Shielder-shaped synthetic relation (rows a1-a5).

PARITY UNPINNED: the reference contains no prover; `north_star`
names arkworks.  ark-groth16 is pinned in no lockfile of the reference; this
file restates the published algorithm of ark-groth16 0.4 as used with
ark-bls12-381 0.4.0 / ark-poly 0.4.2 (shielder/contract/Cargo.lock:195-196,
267-268):
  * generator::generate_parameters_with_qap (LibsnarkReduction::
    instance_map_with_evaluation): QAP evaluated at tau on the radix-2 domain
    of size next_pow2(num_constraints + num_instance_variables), with the
    input-consistency rows a[j] += L_{num_constraints+j}(tau);
  * r1cs_to_qap::LibsnarkReduction::witness_map_from_matrices: 3 iNTT,
    3 coset NTT (g = 7), h = (a*b - c) / Z(g) pointwise, 1 coset iNTT;
  * prover::create_proof_with_assignment: A, B, C as in SURVEY.md row a10;
  * verifier: e(A,B) = e(alpha,beta) e(sum x_i IC_i, gamma) e(C, delta).
Setup randomness (tau, alpha, beta, gamma, delta) and prover randomness (r, s)
are explicit inputs so proofs are reproducible byte-for-byte.
"""
from . import bls12_381 as ec
from .bls12_381 import R, Fq, Fq2
from . import ntt as nt


class R1CS:
    """Rows are lists of (column, coefficient).  Column 0 is the constant 1;
    columns [0, n_pub) are instance variables (n_pub counts the constant)."""

    def __init__(self, n_vars, n_pub, A, B, C):
        self.n_vars, self.n_pub = n_vars, n_pub
        self.A, self.B, self.C = A, B, C
        self.n_constraints = len(A)
        m = self.n_constraints + n_pub
        self.log_n = max(1, (m - 1).bit_length())
        self.N = 1 << self.log_n

    def is_satisfied(self, z):
        ev = lambda row: sum(c * z[j] for j, c in row) % R
        return all(
            ev(a) * ev(b) % R == ev(c) for a, b, c in zip(self.A, self.B, self.C)
        )


# --------------------------------------------------------------------------
# Shielder-shaped synthetic relation
# --------------------------------------------------------------------------
TREE_HEIGHT = 10  # shielder/mocked_zk/src/lib.rs:16 MERKLE_TREE_DEPTH
N_PUB = 7  # 1, op_pub(amount, token, user), new_note_hash, merkle_root, old_nullifier
# variable indices; order follows UpdateNoteInput::new
# (shielder/relations/src/relations/update_note.rs:47-88) with the public
# inputs first in the order update_note_circuit makes them public (:121,:127)
V_ONE, V_AMOUNT, V_TOKEN, V_USER, V_NEW_NOTE_HASH, V_MERKLE_ROOT, V_OLD_NULLIFIER = range(7)
V_NEW_NOTE = 7  # zk_id, trapdoor, nullifier, account_hash (note.rs:25-31)
V_OLD_NOTE = 11  # zk_id, trapdoor, account_hash (nullifier is public)
V_PATH_SHAPE = 14  # TREE_HEIGHT selector bits (merkle_proof.rs:27-34)
V_PATH = 24  # TREE_HEIGHT siblings
V_OP_PRIV_USER = 34
V_OLD_ACCOUNT = 35  # TOKENS_NUMBER = 2 balances
V_CHAIN = 37  # first padding variable
N_FIXED_CONSTRAINTS = TREE_HEIGHT + 2
N_RECHECK = 18


def shielder_r1cs(log_n):
    """Synthetic withdraw-shaped relation with constraints + instance = 2^log_n
    and exactly 2^log_n variables (so every MSM has 2^log_n - 1 terms).

    rows 0..9   : path_shape[i] * (path_shape[i] - 1) = 0      (merkle_proof.rs:53)
    row  10     : (op_priv.user - op_pub.user) * 1 = 0         (mocked_zk ops.rs:47-63)
    row  11     : (new.zk_id - old.zk_id) * 1 = 0              (mocked_zk relations.rs:57-77)
    chain rows  : s_k * s_k = s_{k+1} - s_{k-1}   (stand-in for the Poseidon
                  permutations; binds new_note_hash and merkle_root to the
                  private fields the way a hash output would)
    recheck rows: (2 s_k) * s_k = 2 s_{k+1} - 2 s_{k-1} for the first 18 k
    where s_{-1} = linear combination of the public inputs,
          s_0    = linear combination of all loaded private fields,
          s_mid  = new_note_hash, s_last = merkle_root.
    """
    N = 1 << log_n
    assert N >= 128
    n_vars = N
    n_chain_vars = n_vars - V_CHAIN  # s_1 .. s_K are variables
    K = n_chain_vars
    mid = K // 2
    A, B, C = [], [], []
    for i in range(TREE_HEIGHT):
        A.append([(V_PATH_SHAPE + i, 1)])
        B.append([(V_PATH_SHAPE + i, 1), (V_ONE, R - 1)])
        C.append([])
    A.append([(V_OP_PRIV_USER, 1), (V_USER, R - 1)])
    B.append([(V_ONE, 1)])
    C.append([])
    A.append([(V_NEW_NOTE, 1), (V_OLD_NOTE, R - 1)])
    B.append([(V_ONE, 1)])
    C.append([])

    def s_row(k):
        """linear combination representing s_k, k in [-1, K]."""
        if k == -1:
            return [(V_AMOUNT, 1), (V_TOKEN, 2), (V_USER, 3), (V_OLD_NULLIFIER, 4)]
        if k == 0:
            return [(V_NEW_NOTE + j, j + 1) for j in range(V_CHAIN - V_NEW_NOTE)]
        if k == mid:
            return [(V_NEW_NOTE_HASH, 1)]
        if k == K:
            return [(V_MERKLE_ROOT, 1)]
        return [(V_CHAIN + k - 1, 1)]

    # chain rows k = 0 .. K-1 define s_{k+1}; note s_mid and s_K live in the
    # public block, so variables V_CHAIN+mid-1 and V_CHAIN+K-1 are *free
    # copies* constrained below to equal them (keeps n_vars a power of two).
    for k in range(K):
        A.append(s_row(k))
        B.append(s_row(k))
        C.append(s_row(k + 1) + [(j, (R - c) % R) for j, c in s_row(k - 1)])
    # copy constraints for the two shadowed chain slots
    A.append([(V_CHAIN + mid - 1, 1), (V_NEW_NOTE_HASH, R - 1)])
    B.append([(V_ONE, 1)])
    C.append([])
    A.append([(V_CHAIN + K - 1, 1), (V_MERKLE_ROOT, R - 1)])
    B.append([(V_ONE, 1)])
    C.append([])
    n_re = N - N_PUB - len(A)
    assert n_re == N_RECHECK - 2, n_re
    for k in range(n_re):
        A.append([(j, 2 * c % R) for j, c in s_row(k)])
        B.append(s_row(k))
        C.append(
            [(j, 2 * c % R) for j, c in s_row(k + 1)]
            + [(j, (R - 2 * c) % R) for j, c in s_row(k - 1)]
        )
    r1cs = R1CS(n_vars, N_PUB, A, B, C)
    assert r1cs.N == N and r1cs.n_constraints + N_PUB == N
    return r1cs


def shielder_witness(log_n, seed):
    """Full assignment z for shielder_r1cs(log_n) from a SplitMix64 seed."""
    r1cs_n = 1 << log_n
    rng = ec.SplitMix64(seed)
    z = [0] * r1cs_n
    z[V_ONE] = 1
    z[V_AMOUNT] = rng.next() & 0xFFFFFFFF
    z[V_TOKEN] = rng.fr()
    z[V_USER] = rng.fr()
    z[V_OLD_NULLIFIER] = rng.fr()
    for j in range(4):
        z[V_NEW_NOTE + j] = rng.fr()
    z[V_OLD_NOTE] = z[V_NEW_NOTE]  # same zk_id
    z[V_OLD_NOTE + 1] = rng.fr()
    z[V_OLD_NOTE + 2] = rng.fr()
    for i in range(TREE_HEIGHT):
        z[V_PATH_SHAPE + i] = rng.next() & 1
    for i in range(TREE_HEIGHT):
        z[V_PATH + i] = rng.fr()
    z[V_OP_PRIV_USER] = z[V_USER]
    z[V_OLD_ACCOUNT] = rng.next() & 0xFFFFFFFFFFFF
    z[V_OLD_ACCOUNT + 1] = rng.next() & 0xFFFFFFFFFFFF
    K = r1cs_n - V_CHAIN
    mid = K // 2
    s_prev = (z[V_AMOUNT] + 2 * z[V_TOKEN] + 3 * z[V_USER] + 4 * z[V_OLD_NULLIFIER]) % R
    s_cur = sum((j + 1) * z[V_NEW_NOTE + j] for j in range(V_CHAIN - V_NEW_NOTE)) % R
    for k in range(K):
        s_next = (s_cur * s_cur + s_prev) % R
        z[V_CHAIN + k] = s_next
        s_prev, s_cur = s_cur, s_next
    z[V_NEW_NOTE_HASH] = z[V_CHAIN + mid - 1]
    z[V_MERKLE_ROOT] = z[V_CHAIN + K - 1]
    return z


# --------------------------------------------------------------------------
# QAP helpers
# --------------------------------------------------------------------------
def lagrange_at(log_n, tau):
    """[L_i(tau)] for the radix-2 domain (ark_poly evaluate_all_lagrange_coefficients)."""
    N = 1 << log_n
    w = nt.root_of_unity(log_n)
    zt = (pow(tau, N, R) - 1) % R
    assert zt != 0
    ninv = pow(N, R - 2, R)
    out = []
    wi = 1
    for _ in range(N):
        out.append(zt * ninv % R * wi % R * pow((tau - wi) % R, R - 2, R) % R)
        wi = wi * w % R
    return out


def setup(r1cs, tau, alpha, beta, gamma, delta):
    """ark_groth16::generator::generate_parameters_with_qap, restated."""
    N, nc, npub, nv = r1cs.N, r1cs.n_constraints, r1cs.n_pub, r1cs.n_vars
    L = lagrange_at(r1cs.log_n, tau)
    a = [0] * nv
    b = [0] * nv
    c = [0] * nv
    for j in range(npub):
        a[j] = L[nc + j]
    for i in range(nc):
        for j, v in r1cs.A[i]:
            a[j] = (a[j] + L[i] * v) % R
        for j, v in r1cs.B[i]:
            b[j] = (b[j] + L[i] * v) % R
        for j, v in r1cs.C[i]:
            c[j] = (c[j] + L[i] * v) % R
    zt = (pow(tau, N, R) - 1) % R
    ginv = pow(gamma, R - 2, R)
    dinv = pow(delta, R - 2, R)
    pk = {
        "alpha_g1": ec.g1_mul(alpha),
        "beta_g1": ec.g1_mul(beta),
        "beta_g2": ec.g2_mul(beta),
        "delta_g1": ec.g1_mul(delta),
        "delta_g2": ec.g2_mul(delta),
        "a_query": [ec.g1_mul(v) for v in a],
        "b_g1_query": [ec.g1_mul(v) for v in b],
        "b_g2_query": [ec.g2_mul(v) for v in b],
        "h_query": [ec.g1_mul(pow(tau, i, R) * zt % R * dinv) for i in range(N - 1)],
        "l_query": [
            ec.g1_mul((beta * a[j] + alpha * b[j] + c[j]) % R * dinv)
            for j in range(npub, nv)
        ],
    }
    vk = {
        "alpha_g1": pk["alpha_g1"],
        "beta_g2": pk["beta_g2"],
        "gamma_g2": ec.g2_mul(gamma),
        "delta_g2": pk["delta_g2"],
        "gamma_abc_g1": [
            ec.g1_mul((beta * a[j] + alpha * b[j] + c[j]) % R * ginv)
            for j in range(npub)
        ],
    }
    return pk, vk


def witness_map(r1cs, z):
    """h coefficients (length N), LibsnarkReduction::witness_map_from_matrices."""
    N, nc, npub = r1cs.N, r1cs.n_constraints, r1cs.n_pub
    ev = lambda row: sum(c * z[j] for j, c in row) % R
    a = [ev(r1cs.A[i]) for i in range(nc)] + [0] * (N - nc)
    b = [ev(r1cs.B[i]) for i in range(nc)] + [0] * (N - nc)
    c = [ev(r1cs.C[i]) for i in range(nc)] + [0] * (N - nc)
    for j in range(npub):
        a[nc + j] = z[j]
    a = nt.coset_ntt(nt.ntt(a, inverse=True))
    b = nt.coset_ntt(nt.ntt(b, inverse=True))
    c = nt.coset_ntt(nt.ntt(c, inverse=True))
    zinv = pow((pow(ec.FR_GENERATOR, N, R) - 1) % R, R - 2, R)
    ab = [((x * y - w) % R) * zinv % R for x, y, w in zip(a, b, c)]
    return nt.coset_intt(ab)


def prove(pk, r1cs, z, r, s):
    """ark_groth16::prover::create_proof_with_assignment, restated.
    Returns affine (A in G1, B in G2, C in G1)."""
    npub = r1cs.n_pub
    h = witness_map(r1cs, z)
    assignment = z[1:]
    aux = z[npub:]
    h_acc = ec.msm_naive(Fq, h[: r1cs.N - 1], pk["h_query"])
    l_acc = ec.msm_naive(Fq, aux, pk["l_query"])

    def coeff(F, init, query, vk_param):
        acc = ec.msm_naive(F, assignment, query[1:])
        return ec.pt_sum(F, [init, query[0], acc, vk_param])

    g_a = coeff(Fq, ec.g1_mul(r, pk["delta_g1"]), pk["a_query"], pk["alpha_g1"])
    g1_b = coeff(Fq, ec.g1_mul(s, pk["delta_g1"]), pk["b_g1_query"], pk["beta_g1"])
    g2_b = coeff(Fq2, ec.g2_mul(s, pk["delta_g2"]), pk["b_g2_query"], pk["beta_g2"])
    g_c = ec.pt_sum(
        Fq,
        [
            ec.g1_mul(s, g_a),
            ec.g1_mul(r, g1_b),
            ec.pt_neg(Fq, ec.g1_mul(r * s % R, pk["delta_g1"])),
            l_acc,
            h_acc,
        ],
    )
    return g_a, g2_b, g_c


def proof_to_bytes(proof):
    a, b, c = proof
    return ec.g1_compress(a) + ec.g2_compress(b) + ec.g1_compress(c)


def proof_from_bytes(b):
    assert len(b) == 192
    return ec.g1_decompress(b[:48]), ec.g2_decompress(b[48:144]), ec.g1_decompress(b[144:])


def verify(vk, publics, proof):
    """publics excludes the leading 1.  Pairing-product check."""
    a, b, c = proof
    ic = vk["gamma_abc_g1"]
    assert len(publics) + 1 == len(ic)
    acc = ec.pt_sum(Fq, [ic[0]] + [ec.g1_mul(x, q) for x, q in zip(publics, ic[1:])])
    return ec.pairing_product_is_one(
        [
            (ec.pt_neg(Fq, a), b),
            (vk["alpha_g1"], vk["beta_g2"]),
            (acc, vk["gamma_g2"]),
            (c, vk["delta_g2"]),
        ]
    )

"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Poseidon (x^5) over a prime field, with the parameters the reference's relations fix
(shielder/relations/src/lib.rs:17-26: T_WIDTH = 5, RATE = 4, R_F = 8, R_P = 56) and the
sponge framing of `PoseidonHasher::hash_fix_len_array`, the only hashing call the reference
makes (update_note.rs:100,131, update_account.rs:62, merkle_proof.rs:56).

The arithmetic lives in third-party crates that are NOT in /root/reference:
halo2-base 0.4.1 (shielder/Cargo.lock:414-416) and poseidon 0.2.0 (zemse/pse-poseidon,
shielder/Cargo.lock:1029-1031).  Their `OptimizedPoseidonSpec::new::<R_F, R_P, 0>()` derives the
round constants and the MDS matrix with the Grain LFSR of the Poseidon paper (field tag 1, s-box
tag 0, n = NUM_BITS, t, R_F, R_P; 160 bits discarded; bits taken in pairs; round constants by
rejection sampling; then 2t elements *without* rejection for the Cauchy matrix 1/(x_i + y_j)).
That published procedure is restated here.

PIN STATUS
 * constant generation + MDS + permutation: PINNED by published known-answer vectors of the same
   procedure over BN254 Fr (circomlib's poseidon constants/tests, quoted from memory and
   reproduced exactly by this code; see tests/test_cpu_oracle.py):
     first round constant (t=2, 8/56) = 0x09c46e9e...d7a7,  (t=3, 8/57) = 0x0ee9a592...8e6e,
     permute([0,1,2])[0]     (t=3, 8/57) = 0x115cc0f5...189a,
     permute([0,1,2,3,4])[0] (t=5, 8/60) = 0x299c867d...0465.
 * the optimized form used by halo2-base (pre-sparse MDS, sparse partial-round matrices) computes
   the same permutation as the plain ARK -> S-box -> MDS rounds below (that equivalence is the
   point of the optimisation); only the plain form is restated.
 * sponge framing (initial state [2^64, 0, ...], "+1" padding at position len+1, an extra
   permutation when len % RATE == 0, output = state[1]): restated from pse-poseidon /
   halo2-base, PARITY UNPINNED — the reference's tests hold no Poseidon vector
   (`relations` has zero tests) and its field is BN254 Fr, not BLS12-381 Fr.
"""
from functools import lru_cache

from .bls12_381 import R as BLS_FR

BN254_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617

T_WIDTH, RATE, R_F, R_P = 5, 4, 8, 56  # shielder/relations/src/lib.rs:17-26
TREE_HEIGHT = 10  # shielder/mocked_zk/src/lib.rs:16

FIELDS = {"bls12_381_fr": (BLS_FR, 255), "bn254_fr": (BN254_FR, 254)}


class Grain:
    """80-bit LFSR b[i+80] = b[i+62]^b[i+51]^b[i+38]^b[i+23]^b[i+13]^b[i] (Poseidon paper, app. F)."""

    def __init__(self, nbits, t, r_f, r_p, sbox_tag=0, field_tag=1):
        bits = []
        for value, width in ((field_tag, 2), (sbox_tag, 4), (nbits, 12), (t, 12), (r_f, 10), (r_p, 10), ((1 << 30) - 1, 30)):
            bits.extend((value >> (width - 1 - i)) & 1 for i in range(width))
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._clock()

    def _clock(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(b)
        return b

    def bit(self):
        # bits are consumed in pairs: a leading 1 emits the second bit, a leading 0 drops it
        while True:
            first, second = self._clock(), self._clock()
            if first:
                return second

    def integer(self, nbits):
        v = 0
        for _ in range(nbits):  # most significant bit first
            v = (v << 1) | self.bit()
        return v

    def field_element(self, p, nbits):
        while True:
            v = self.integer(nbits)
            if v < p:
                return v


@lru_cache(maxsize=None)
def spec(field="bls12_381_fr", t=T_WIDTH, r_f=R_F, r_p=R_P):
    """(round_constants[r_f + r_p][t], mds[t][t]) for the field."""
    p, nbits = FIELDS[field]
    g = Grain(nbits, t, r_f, r_p)
    rc = tuple(tuple(g.field_element(p, nbits) for _ in range(t)) for _ in range(r_f + r_p))
    while True:
        v = [g.integer(nbits) % p for _ in range(2 * t)]
        if len(set(v)) == 2 * t:
            break
    xs, ys = v[:t], v[t:]
    mds = tuple(tuple(pow(xs[i] + ys[j], -1, p) for j in range(t)) for i in range(t))
    return rc, mds


def permute(state, field="bls12_381_fr", r_f=R_F, r_p=R_P):
    p, _ = FIELDS[field]
    t = len(state)
    rc, mds = spec(field, t, r_f, r_p)
    st = list(state)
    for r in range(r_f + r_p):
        st = [(a + c) % p for a, c in zip(st, rc[r])]
        if r < r_f // 2 or r >= r_f // 2 + r_p:
            st = [pow(a, 5, p) for a in st]
        else:
            st[0] = pow(st[0], 5, p)
        st = [sum(mds[i][j] * st[j] for j in range(t)) % p for i in range(t)]
    return st


def hash_fix_len(inputs, field="bls12_381_fr"):
    """PoseidonHasher::<F, 5, 4>::hash_fix_len_array (halo2-base 0.4.1, not in tree)."""
    p, _ = FIELDS[field]
    st = [(1 << 64) % p] + [0] * RATE
    inputs = [x % p for x in inputs]
    chunks = [inputs[i : i + RATE] for i in range(0, len(inputs), RATE)]
    if len(inputs) % RATE == 0:
        chunks.append([])
    for chunk in chunks:
        for i, x in enumerate(chunk):
            st[1 + i] = (st[1 + i] + x) % p
        if len(chunk) < RATE:
            st[1 + len(chunk)] = (st[1 + len(chunk)] + 1) % p
        st = permute(st, field)
    return st[1]


def note_hash(zk_id, trapdoor, nullifier, account_hash, field="bls12_381_fr"):
    """verify_note_circuit (update_note.rs:91-103): hash of Note::clone_to_vec (note.rs:32-36)."""
    return hash_fix_len([zk_id, trapdoor, nullifier, account_hash], field)


def merkle_root(leaf, path_shape, path, field="bls12_381_fr"):
    """CircuitMerkleProof::verify (merkle_proof.rs:38-61): selector = is_zero(shape);
    left = select(sibling, current, selector); right = select(current, sibling, selector)."""
    cur = leaf
    for shape, sibling in zip(path_shape, path):
        left, right = (sibling, cur) if shape == 0 else (cur, sibling)
        cur = hash_fix_len([left, right], field)
    return cur


def merkle_tree(leaves, field="bls12_381_fr"):
    """All levels of the binary Poseidon tree over `leaves` (power of two), leaves first."""
    levels = [list(leaves)]
    while len(levels[-1]) > 1:
        cur = levels[-1]
        levels.append([hash_fix_len([cur[2 * i], cur[2 * i + 1]], field) for i in range(len(cur) // 2)])
    return levels

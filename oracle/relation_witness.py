"""TEST INFRASTRUCTURE (oracle/): an assignment for the update_note / create_note relations that does NOT come from the
product's witness generator.

Two independent ingredients:
  * the LOADED values -- what `UpdateNoteInput::new` loads, in its order
    (/root/reference/shielder/relations/src/relations/update_note.rs:47-88: op_pub, new_note_hash, merkle_root, new note,
    old note, Merkle path shape + siblings (merkle_proof.rs:27-34), op_priv, old account; the public ones first, in the order
    `update_note_circuit` makes them public, :121,:127) -- with every hash among them computed by oracle/poseidon.py
    (note hashes note.rs:25-31, the Merkle fold merkle_proof.rs:38-61, the account hashes update_account.rs:68-95);
  * a generic R1CS SOLVER over the matrices the relation exports: every remaining variable is defined by the constraint that
    introduces it, so walking the rows in order determines the whole assignment.  Three row shapes introduce variables:
      a * b = c with ONE unknown in c          products, x^5 chains, selects, the padding chain        -> solved linearly
      x * inv = 1 - out ; x * out = 0          GateInstructions::is_zero (two unknowns, consecutive)  -> out = [x == 0]
      b * (b - 1) = 0 ... ; sum 2^i b_i = x    the 128-bit range check of a balance                    -> bits of x
    Rows without unknowns are CHECKED; variables no row mentions are 0.

If the constraint system did not compute the oracle's Poseidon, the loaded hashes would contradict the solved ones and
`solve` raises -- the relation and the prover are then exercised by inputs neither of which the product produced.
Only tests/ may import this module."""
from .bls12_381 import R
from . import poseidon as ps


def update_note_loaded(op_kind, amount, token, user, new_note, old_note, path_shape, path, priv_user, account):
    """z[0 .. n_loaded) of update_note: new_note / old_note = (zk_id, trapdoor, nullifier), account = (token_0, balance_0,
    token_1, balance_1) BEFORE the operation; op_kind 0 = deposit, 1 = withdraw (mocked_zk/src/ops.rs:6-25)."""
    t0, b0, t1, b1 = account
    slot = 0 if t0 == token else 1
    nb = [b0, b1]
    nb[slot] = (nb[slot] + amount) % R if op_kind == 0 else (nb[slot] - amount) % R
    old_acc = ps.hash_fix_len([t0, b0, t1, b1])
    new_acc = ps.hash_fix_len([t0, nb[0], t1, nb[1]])
    new_hash = ps.note_hash(new_note[0], new_note[1], new_note[2], new_acc)
    old_hash = ps.note_hash(old_note[0], old_note[1], old_note[2], old_acc)
    root = ps.merkle_root(old_hash, list(path_shape), list(path))
    return ([1, amount, token, user, new_hash, root, old_note[2]]
            + [new_note[0], new_note[1], new_note[2], new_acc]
            + [old_note[0], old_note[1], old_acc]
            + list(path_shape) + list(path) + [priv_user] + [t0, b0, t1, b1])


def create_note_loaded(tokens, note):
    """z[0 .. n_loaded) of the creation relation (update_note.rs:91-103 + update_account.rs:52-65): publics h_note_new,
    token_0, token_1; then the note (zk_id, trapdoor, nullifier) and its account hash over zero balances."""
    acc = ps.hash_fix_len([tokens[0], 0, tokens[1], 0])
    return [1, ps.note_hash(note[0], note[1], note[2], acc), tokens[0], tokens[1]], acc


def solve(n_vars, A, B, C, known):
    """Full assignment from `known` (dict or list prefix of loaded values) and the rows (lists of (column, coefficient))."""
    z = [None] * n_vars
    if isinstance(known, dict):
        for k, v in known.items():
            z[k] = v % R
    else:
        for k, v in enumerate(known):
            z[k] = v % R
    z[0] = 1

    def ev(row):  # (value of the known part, [(col, coef)] of the unknown part)
        s, unk = 0, []
        for j, c in row:
            if z[j] is None:
                unk.append((j, c))
            else:
                s += c * z[j]
        return s % R, unk

    pending_bits = []  # variables constrained to {0, 1} whose value the next linear row fixes
    i, n = 0, len(A)
    while i < n:
        (a, ua), (b, ub), (c, uc) = ev(A[i]), ev(B[i]), ev(C[i])
        if not ua and not ub and not uc:
            if a * b % R != c:
                raise ValueError("constraint %d contradicts the loaded values" % i)
        elif not ua and not ub and len(uc) == 1:
            j, k = uc[0]
            z[j] = (a * b - c) * pow(k, -1, R) % R
        elif not ua and len(ub) == 1 and len(uc) == 1 and i + 1 < n:
            # is_zero: x * inv = 1 - out, then x * out = 0
            (jinv, kinv), (jout, kout) = ub[0], uc[0]
            nb_row, nc_row = B[i + 1], C[i + 1]
            if not (kinv == 1 and kout == R - 1 and c == 1 and b == 0 and list(nb_row) == [(jout, 1)] and not nc_row and list(A[i + 1]) == list(A[i])):
                raise ValueError("constraint %d: two unknowns in a shape that is not is_zero" % i)
            z[jout] = 1 if a == 0 else 0
            z[jinv] = 0 if a == 0 else pow(a, -1, R)
        elif len(ua) == 1 and len(ub) == 1 and ua[0][0] == ub[0][0] and not uc and a == 0 and b == R - 1 and c == 0 and ua[0][1] == 1 and ub[0][1] == 1:
            pending_bits.append(ua[0][0])  # b * (b - 1) = 0
        elif pending_bits and not ub and not uc and b == 1 and c == 0 and {j for j, _ in ua} == set(pending_bits):
            # (sum 2^i b_i - x) * 1 = 0: the bits of x, weights read from the row
            x = (-a) % R
            weights = sorted(((k, j) for j, k in ua), reverse=True)
            for k, j in weights:
                bit = 1 if x >= k else 0
                z[j] = bit
                x -= bit * k
            if x != 0:
                raise ValueError("constraint %d: value does not fit its range check" % i)
            pending_bits = []
        else:
            raise ValueError("constraint %d: %d + %d + %d unknowns in an unexpected shape" % (i, len(ua), len(ub), len(uc)))
        i += 1
    if pending_bits:
        raise ValueError("bit variables without their sum row")
    return [0 if v is None else v for v in z]

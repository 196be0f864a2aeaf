// zkmi — R1CS container and the Shielder-shaped synthetic relation (host code).
//
// Reference anchors (what the reference *does* define about the relation):
//   witness load order  UpdateNoteInput::new   shielder/relations/src/relations/update_note.rs:47-88
//   public-input order  update_note_circuit    update_note.rs:121,127
//                       (op_pub.., new_note_hash, merkle_root, old_note.nullifier)
//   note fields         shielder/relations/src/note.rs:25-31
//   merkle path         shielder/relations/src/merkle_proof.rs:27-34 (TREE_HEIGHT shape bits + siblings)
//   TREE_HEIGHT = 10    shielder/mocked_zk/src/lib.rs:16
// The Poseidon permutations of the real circuit need halo2-base's generated
// constants, which are not in the tree; they are replaced by a
// multiplication chain of the same role (binds the public hash outputs to the
// private fields) padded to constraints + instance variables = 2^log_n.
#include <string.h>
#include <new>
#include "ctx.hpp"
#include "r1cs.hpp"

namespace zkmi {

namespace {
enum {
  V_ONE = 0, V_AMOUNT, V_TOKEN, V_USER, V_NEW_NOTE_HASH, V_MERKLE_ROOT, V_OLD_NULLIFIER,
  N_PUB = 7,
  V_NEW_NOTE = 7,      // zk_id, trapdoor, nullifier, account_hash
  V_OLD_NOTE = 11,     // zk_id, trapdoor, account_hash (nullifier is public)
  V_PATH_SHAPE = 14,   // TREE_HEIGHT selector bits
  V_PATH = 24,         // TREE_HEIGHT siblings
  V_OP_PRIV_USER = 34,
  V_OLD_ACCOUNT = 35,  // TOKENS_NUMBER balances
  V_CHAIN = 37,
  TREE_HEIGHT = 10
};

Fr fr_u64(uint64_t v) {
  Fr a = Fr::zero();
  a.l[0] = (uint32_t)v;
  a.l[1] = (uint32_t)(v >> 32);
  return a.to_mont();
}

struct Term {
  uint32_t col;
  Fr coef;
};
typedef std::vector<Term> Row;

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform in [0, r) by rejection on 255-bit candidates; Montgomery form out
  Fr fr() {
    for (;;) {
      uint8_t b[32];
      for (int i = 0; i < 4; i++) {
        uint64_t v = next();
        memcpy(b + 8 * i, &v, 8);
      }
      b[31] &= 0x7f;
      Fr out;
      if (fr_from_wire(b, &out)) return out;
    }
  }
};

void push_row(zkmi_r1cs::Csr& m, const Row& row) {
  for (const Term& t : row) {
    m.col.push_back(t.col);
    m.val.push_back(t.coef);
  }
  m.rowptr.push_back((uint32_t)m.col.size());
}
}  // namespace

void r1cs_finish_shape(zkmi_r1cs* r) {
  uint64_t m = (uint64_t)r->n_constraints + r->n_pub;
  uint32_t lg = 1;
  while ((1ull << lg) < m) lg++;
  r->log_n = lg;
}

#ifdef ZKMI_TESTING  // the hash-free chain stand-in of the first builds: golden fixtures only (libzkmi_exp.so)
zkmi_r1cs* build_shielder_r1cs(uint32_t log_n) {
  if (log_n < 7 || log_n > 26) return nullptr;
  const uint32_t N = 1u << log_n;
  zkmi_r1cs* r = new (std::nothrow) zkmi_r1cs();
  if (!r) return nullptr;
  r->n_vars = N;
  r->n_pub = N_PUB;
  const uint32_t K = N - V_CHAIN;  // chain variables s_1..s_K
  const uint32_t mid = K / 2;
  for (int i = 0; i < 3; i++) r->m[i].rowptr.push_back(0);
  const Fr one = Fr::one(), minus_one = Fr::one().neg();
  const Fr two = fr_u64(2), minus_two = fr_u64(2).neg();
  auto add = [&](const Row& a, const Row& b, const Row& c) {
    push_row(r->m[0], a);
    push_row(r->m[1], b);
    push_row(r->m[2], c);
  };
  for (int i = 0; i < TREE_HEIGHT; i++)
    add({{(uint32_t)(V_PATH_SHAPE + i), one}}, {{(uint32_t)(V_PATH_SHAPE + i), one}, {V_ONE, minus_one}}, {});
  add({{V_OP_PRIV_USER, one}, {V_USER, minus_one}}, {{V_ONE, one}}, {});
  add({{V_NEW_NOTE, one}, {V_OLD_NOTE, minus_one}}, {{V_ONE, one}}, {});
  // s_k as a linear combination, k in [-1, K]
  auto s_row = [&](int64_t k, const Fr& scale) -> Row {
    Row row;
    if (k == -1) {
      row = {{V_AMOUNT, fr_u64(1)}, {V_TOKEN, fr_u64(2)}, {V_USER, fr_u64(3)}, {V_OLD_NULLIFIER, fr_u64(4)}};
    } else if (k == 0) {
      for (uint32_t j = 0; j < V_CHAIN - V_NEW_NOTE; j++) row.push_back({V_NEW_NOTE + j, fr_u64(j + 1)});
    } else if ((uint32_t)k == mid) {
      row = {{V_NEW_NOTE_HASH, one}};
    } else if ((uint32_t)k == K) {
      row = {{V_MERKLE_ROOT, one}};
    } else {
      row = {{(uint32_t)(V_CHAIN + k - 1), one}};
    }
    for (Term& t : row) t.coef = t.coef * scale;
    return row;
  };
  auto c_row = [&](int64_t k, const Fr& pos, const Fr& negs) -> Row {
    Row c = s_row(k + 1, pos);
    Row d = s_row(k - 1, negs);
    c.insert(c.end(), d.begin(), d.end());
    return c;
  };
  for (uint32_t k = 0; k < K; k++) add(s_row(k, one), s_row(k, one), c_row(k, one, minus_one));
  add({{V_CHAIN + mid - 1, one}, {V_NEW_NOTE_HASH, minus_one}}, {{V_ONE, one}}, {});
  add({{V_CHAIN + K - 1, one}, {V_MERKLE_ROOT, minus_one}}, {{V_ONE, one}}, {});
  const uint32_t n_re = N - N_PUB - (uint32_t)(r->m[0].rowptr.size() - 1);
  for (uint32_t k = 0; k < n_re; k++) add(s_row(k, two), s_row(k, one), c_row(k, two, minus_two));
  r->n_constraints = (uint32_t)(r->m[0].rowptr.size() - 1);
  r1cs_finish_shape(r);
  return r;
}

// chain part of the assignment: s_{k+1} = s_k^2 + s_{k-1}, then the two public "hash outputs"
static void fill_chain(uint32_t N, std::vector<Fr>& z) {
  const uint32_t K = N - V_CHAIN, mid = K / 2;
  Fr s_prev = z[V_AMOUNT] + fr_u64(2) * z[V_TOKEN] + fr_u64(3) * z[V_USER] + fr_u64(4) * z[V_OLD_NULLIFIER];
  Fr s_cur = Fr::zero();
  for (uint32_t j = 0; j < V_CHAIN - V_NEW_NOTE; j++) s_cur = s_cur + fr_u64(j + 1) * z[V_NEW_NOTE + j];
  for (uint32_t k = 0; k < K; k++) {
    Fr s_next = s_cur.sqr() + s_prev;
    z[V_CHAIN + k] = s_next;
    s_prev = s_cur;
    s_cur = s_next;
  }
  z[V_NEW_NOTE_HASH] = z[V_CHAIN + mid - 1];
  z[V_MERKLE_ROOT] = z[V_CHAIN + K - 1];
}

void build_shielder_witness(uint32_t log_n, uint64_t seed, std::vector<Fr>* zp) {
  const uint32_t N = 1u << log_n;
  std::vector<Fr>& z = *zp;
  z.assign(N, Fr::zero());
  SplitMix64 rng{seed};
  z[V_ONE] = Fr::one();
  z[V_AMOUNT] = fr_u64(rng.next() & 0xFFFFFFFFull);
  z[V_TOKEN] = rng.fr();
  z[V_USER] = rng.fr();
  z[V_OLD_NULLIFIER] = rng.fr();
  for (int j = 0; j < 4; j++) z[V_NEW_NOTE + j] = rng.fr();
  z[V_OLD_NOTE] = z[V_NEW_NOTE];
  z[V_OLD_NOTE + 1] = rng.fr();
  z[V_OLD_NOTE + 2] = rng.fr();
  for (int i = 0; i < TREE_HEIGHT; i++) z[V_PATH_SHAPE + i] = fr_u64(rng.next() & 1);
  for (int i = 0; i < TREE_HEIGHT; i++) z[V_PATH + i] = rng.fr();
  z[V_OP_PRIV_USER] = z[V_USER];
  z[V_OLD_ACCOUNT] = fr_u64(rng.next() & 0xFFFFFFFFFFFFull);
  z[V_OLD_ACCOUNT + 1] = fr_u64(rng.next() & 0xFFFFFFFFFFFFull);
  fill_chain(N, z);
}

// Row a1: the assignment from the relation's semantic inputs, loaded in the order of
// UpdateNoteInput::new (shielder/relations/src/relations/update_note.rs:47-88).
bool build_shielder_witness_from_input(uint32_t log_n, const zkmi_update_note_input& in, std::vector<Fr>* zp) {
  const uint32_t N = 1u << log_n;
  std::vector<Fr>& z = *zp;
  z.assign(N, Fr::zero());
  z[V_ONE] = Fr::one();
  bool ok = fr_from_wire(in.amount.bytes, &z[V_AMOUNT]) && fr_from_wire(in.token.bytes, &z[V_TOKEN]) &&
            fr_from_wire(in.user.bytes, &z[V_USER]) && fr_from_wire(in.old_nullifier.bytes, &z[V_OLD_NULLIFIER]);
  for (int j = 0; j < 4 && ok; j++) ok = fr_from_wire(in.new_note[j].bytes, &z[V_NEW_NOTE + j]);
  z[V_OLD_NOTE] = z[V_NEW_NOTE];  // same zk_id (mocked_zk relations.rs:57-77 keeps the id)
  ok = ok && fr_from_wire(in.old_trapdoor.bytes, &z[V_OLD_NOTE + 1]) &&
       fr_from_wire(in.old_account_hash.bytes, &z[V_OLD_NOTE + 2]);
  for (int i = 0; i < TREE_HEIGHT && ok; i++) {
    if (in.path_shape[i] > 1) ok = false;
    z[V_PATH_SHAPE + i] = fr_u64(in.path_shape[i]);
    ok = ok && fr_from_wire(in.path[i].bytes, &z[V_PATH + i]);
  }
  z[V_OP_PRIV_USER] = z[V_USER];  // Operation::combine requires equal users (ops.rs:47-63)
  for (int i = 0; i < 2 && ok; i++) ok = fr_from_wire(in.old_account[i].bytes, &z[V_OLD_ACCOUNT + i]);
  if (!ok) return false;
  fill_chain(N, z);
  return true;
}
#endif  // ZKMI_TESTING

static Fr eval_row(const zkmi_r1cs::Csr& m, uint32_t i, const std::vector<Fr>& z) {
  Fr acc = Fr::zero();
  for (uint32_t k = m.rowptr[i]; k < m.rowptr[i + 1]; k++) acc = acc + m.val[k] * z[m.col[k]];
  return acc;
}

bool r1cs_satisfied(const zkmi_r1cs& r, const std::vector<Fr>& z) {
  for (uint32_t i = 0; i < r.n_constraints; i++)
    if (eval_row(r.m[0], i, z) * eval_row(r.m[1], i, z) != eval_row(r.m[2], i, z)) return false;
  return true;
}

}  // namespace zkmi

using namespace zkmi;

uint64_t zkmi_layout_r1cs() { return sizeof(zkmi_r1cs); }  // capi.hip zkmi_abi_layout_probe

extern "C" {

int32_t zkmi_r1cs_create(uint32_t n_vars, uint32_t n_pub, uint32_t n_constraints, const uint32_t* a_rowptr,
                         const uint32_t* a_col, const uint8_t* a_val, const uint32_t* b_rowptr, const uint32_t* b_col,
                         const uint8_t* b_val, const uint32_t* c_rowptr, const uint32_t* c_col, const uint8_t* c_val,
                         zkmi_r1cs** out) {
  if (!out || !a_rowptr || !b_rowptr || !c_rowptr || n_pub == 0 || n_pub > n_vars) return ZKMI_ERR_BAD_ARG;
  // domain = next_pow2(constraints + instance variables) must be one the NTT supports (2^26)
  if ((uint64_t)n_constraints + n_pub > (1ull << 26) || n_vars > (1u << 26)) return ZKMI_ERR_BAD_ARG;
  {
    // CSR sanity before anything is copied: row pointers start at 0 and never decrease
    const uint32_t* rps[3] = {a_rowptr, b_rowptr, c_rowptr};
    for (const uint32_t* rp : rps) {
      if (rp[0] != 0) return ZKMI_ERR_BAD_ARG;
      for (uint32_t i = 0; i < n_constraints; i++)
        if (rp[i] > rp[i + 1]) return ZKMI_ERR_BAD_ARG;
    }
  }
  zkmi_r1cs* r = new (std::nothrow) zkmi_r1cs();
  if (!r) return ZKMI_ERR_BAD_ARG;
  r->n_vars = n_vars;
  r->n_pub = n_pub;
  r->n_constraints = n_constraints;
  const uint32_t* rp[3] = {a_rowptr, b_rowptr, c_rowptr};
  const uint32_t* cl[3] = {a_col, b_col, c_col};
  const uint8_t* vl[3] = {a_val, b_val, c_val};
  for (int m = 0; m < 3; m++) {
    r->m[m].rowptr.assign(rp[m], rp[m] + n_constraints + 1);
    const uint32_t nnz = rp[m][n_constraints];
    if (rp[m][0] != 0 || (nnz && (!cl[m] || !vl[m]))) {
      delete r;
      return ZKMI_ERR_BAD_ARG;
    }
    r->m[m].col.assign(cl[m], cl[m] + nnz);
    r->m[m].val.resize(nnz);
    for (uint32_t k = 0; k < nnz; k++) {
      if (cl[m][k] >= n_vars) {
        delete r;
        return ZKMI_ERR_BAD_ARG;
      }
      if (!fr_from_wire(vl[m] + 32ull * k, &r->m[m].val[k])) {
        delete r;
        return ZKMI_ERR_NON_CANONICAL;
      }
    }
  }
  r1cs_finish_shape(r);
  *out = r;
  return ZKMI_OK;
}

int32_t zkmi_r1cs_free(zkmi_r1cs* r) {
  if (!r) return ZKMI_ERR_BAD_ARG;
  delete r;
  return ZKMI_OK;
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_shielder_r1cs(uint32_t log_n, zkmi_r1cs** out) {
  if (!out) return ZKMI_ERR_BAD_ARG;
  *out = build_shielder_r1cs(log_n);
  return *out ? ZKMI_OK : ZKMI_ERR_BAD_ARG;
}
#endif  // ZKMI_TESTING

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_shielder_witness(uint32_t log_n, uint64_t seed, uint8_t* out_z) {
  if (!out_z || log_n < 7 || log_n > 26) return ZKMI_ERR_BAD_ARG;
  std::vector<Fr> z;
  build_shielder_witness(log_n, seed, &z);
  for (size_t i = 0; i < z.size(); i++) fr_to_wire(z[i], out_z + 32 * i);
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_shielder_witness_from_input(uint32_t log_n, const zkmi_update_note_input* in, uint8_t* out_z) {
  if (!in || !out_z || log_n < 7 || log_n > 26) return ZKMI_ERR_BAD_ARG;
  std::vector<Fr> z;
  if (!build_shielder_witness_from_input(log_n, *in, &z)) return ZKMI_ERR_NON_CANONICAL;
  for (size_t i = 0; i < z.size(); i++) fr_to_wire(z[i], out_z + 32 * i);
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

// value mod r: SHA-256 outputs used as scalars by the mock exceed r
int32_t zkmi_fr_reduce(const uint8_t in[32], uint8_t out[32]) {
  if (!in || !out) return ZKMI_ERR_BAD_ARG;
  uint32_t w[8];
  memcpy(w, in, 32);
  for (int round = 0; round < 4; round++) {  // 2^256 / r < 3: at most two subtractions
    bool ge = true;
    for (int i = 7; i >= 0; i--)
      if (w[i] != FrParams::MOD[i]) {
        ge = w[i] > FrParams::MOD[i];
        break;
      }
    if (!ge) break;
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
      const uint64_t d = (uint64_t)w[i] - FrParams::MOD[i] - borrow;
      w[i] = (uint32_t)d;
      borrow = (d >> 63) & 1;
    }
  }
  memcpy(out, w, 32);
  return ZKMI_OK;
}

int32_t zkmi_r1cs_shape(const zkmi_r1cs* r, uint32_t* n_vars, uint32_t* n_pub, uint32_t* n_constraints,
                        uint32_t* log_n) {
  if (!r) return ZKMI_ERR_BAD_ARG;
  if (n_vars) *n_vars = r->n_vars;
  if (n_pub) *n_pub = r->n_pub;
  if (n_constraints) *n_constraints = r->n_constraints;
  if (log_n) *log_n = r->log_n;
  return ZKMI_OK;
}

int32_t zkmi_r1cs_export(const zkmi_r1cs* r, int32_t m, uint32_t* rowptr, uint32_t* col, uint8_t* val,
                         uint64_t* nnz) {
  if (!r || m < 0 || m > 2) return ZKMI_ERR_BAD_ARG;
  const zkmi_r1cs::Csr& c = r->m[m];
  if (nnz) *nnz = c.col.size();
  if (rowptr) memcpy(rowptr, c.rowptr.data(), sizeof(uint32_t) * c.rowptr.size());
  if (col) memcpy(col, c.col.data(), sizeof(uint32_t) * c.col.size());
  if (val)
    for (size_t k = 0; k < c.val.size(); k++) fr_to_wire(c.val[k], val + 32 * k);
  return ZKMI_OK;
}

int32_t zkmi_r1cs_is_satisfied(const zkmi_r1cs* r, const uint8_t* z) {
  if (!r || !z) return ZKMI_ERR_BAD_ARG;
  std::vector<Fr> zm(r->n_vars);
  for (uint32_t i = 0; i < r->n_vars; i++)
    if (!fr_from_wire(z + 32ull * i, &zm[i])) return ZKMI_ERR_NON_CANONICAL;
  return r1cs_satisfied(*r, zm) ? ZKMI_OK : ZKMI_ERR_UNSATISFIED;
}

}  // extern "C"

// zkmi — the reference's update_note relation as an R1CS with real Poseidon hashing (host code).
// SURVEY.md §8a rows a1-a5, §8f-1.
//
// What is mirrored, statement by statement:
//   UpdateNoteInput::new      shielder/relations/src/relations/update_note.rs:47-88   (load order)
//   update_note_circuit       update_note.rs:106-149  (publics: op_pub || new_note_hash ||
//                             merkle_root || old_note.nullifier; note hashes; Merkle proof;
//                             Operation::combine; update_account_circuit)
//   verify_note_circuit       update_note.rs:91-103
//   CircuitMerkleProof::verify merkle_proof.rs:38-61  (is_zero, two selects, Poseidon 2 -> 1)
//   update_account_circuit    update_account.rs:68-95 (old/new account hash checks around update)
// The reference leaves Account / Operation generic (relations/src/account.rs, operation.rs); the
// concrete instance used here is the one its mock defines: two (token, balance) slots, deposit /
// withdraw of `amount` on the slot whose token equals op_pub.token, u128 balances, users of
// op_pub and op_priv equal (mocked_zk/src/account.rs:37-82, ops.rs:47-63).
//
// The reference arithmetises with halo2 (PLONKish); the Groth16 path needs R1CS, so every gate
// becomes rank-1 constraints here: x^5 = 3 products, is_zero = 2, select = 1, equality = 1,
// 128-bit range = 129.  The relation proper has a few thousand constraints; the remainder up to
// constraints + publics = 2^log_n is the multiplication chain SURVEY.md §8d prescribes
// (s_{k+1} = s_k^2 + s_{k-1}, seeded from the loaded inputs).
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <utility>
#include "ctx.hpp"
#include "poseidon.hpp"
#include "r1cs.hpp"
#include "relation_values.hpp"

namespace zkmi {
namespace {

// TREE_HEIGHT is a const generic of the reference's relation (merkle_proof.rs:11); here it is the
// run-time field zkmi_note_update::tree_height (0 = ZKMI_MERKLE_TREE_DEPTH, mocked_zk/src/lib.rs:16)
constexpr int BALANCE_BITS = 128;
constexpr uint32_t N_PUB = 7;  // 1, amount, token, user, new_note_hash, merkle_root, old_nullifier

struct Term {
  uint32_t col;
  Fr coef;
};
struct LC {
  std::vector<Term> t;
  Fr v = Fr::zero();
};

Fr fr_small(uint64_t x) {
  Fr a = Fr::zero();
  a.l[0] = (uint32_t)x;
  a.l[1] = (uint32_t)(x >> 32);
  return a.to_mont();
}

struct PoseidonFr {
  Fr rc[POS_ROUNDS * POS_T], mds[POS_T * POS_T], cap;
  PoseidonFr() {
    const uint8_t* r = poseidon_rc_canonical(ZKMI_FIELD_BLS12_381_FR);
    const uint8_t* m = poseidon_mds_canonical(ZKMI_FIELD_BLS12_381_FR);
    for (int i = 0; i < POS_ROUNDS * POS_T; i++) fr_from_wire(r + 32 * i, &rc[i]);
    for (int i = 0; i < POS_T * POS_T; i++) fr_from_wire(m + 32 * i, &mds[i]);
    uint8_t c[32] = {0};
    c[8] = 1;  // 2^64
    fr_from_wire(c, &cap);
  }
};
const PoseidonFr& pos_fr() {
  static const PoseidonFr p;
  return p;
}

class Builder {
 public:
  zkmi_r1cs* r;          // constraints are recorded when non-null
  std::vector<Fr> z;     // assignment, z[0] = 1
  explicit Builder(zkmi_r1cs* out) : r(out) {
    z.push_back(Fr::one());
    if (r)
      for (int i = 0; i < 3; i++) r->m[i].rowptr.assign(1, 0u);
  }
  uint32_t n_constraints = 0;

  LC var(const Fr& value) {
    z.push_back(value);
    LC l;
    l.t.push_back({(uint32_t)z.size() - 1, Fr::one()});
    l.v = value;
    return l;
  }
  static LC constant(const Fr& c) {
    LC l;
    if (!c.is_zero()) l.t.push_back({0u, c});
    l.v = c;
    return l;
  }
  // canonical form: sorted by column, equal columns merged, zero coefficients dropped
  static void compact(LC& a) {
    std::sort(a.t.begin(), a.t.end(), [](const Term& x, const Term& y) { return x.col < y.col; });
    size_t w = 0;
    for (size_t i = 0; i < a.t.size();) {
      Term acc = a.t[i];
      size_t j = i + 1;
      for (; j < a.t.size() && a.t[j].col == acc.col; j++) acc.coef = acc.coef + a.t[j].coef;
      if (!acc.coef.is_zero()) a.t[w++] = acc;
      i = j;
    }
    a.t.resize(w);
  }
  static void add_scaled(LC& dst, const LC& src, const Fr& k) {
    for (const Term& t : src.t) dst.t.push_back({t.col, t.coef * k});
    dst.v = dst.v + src.v * k;
  }
  static LC add(const LC& a, const LC& b) {
    LC o = a;
    add_scaled(o, b, Fr::one());
    compact(o);
    return o;
  }
  static LC sub(const LC& a, const LC& b) {
    LC o = a;
    add_scaled(o, b, Fr::one().neg());
    compact(o);
    return o;
  }
  void enforce(const LC& a, const LC& b, const LC& c) {
    n_constraints++;
    if (!r) return;
    const LC* rows[3] = {&a, &b, &c};
    for (int m = 0; m < 3; m++) {
      for (const Term& t : rows[m]->t) {
        r->m[m].col.push_back(t.col);
        r->m[m].val.push_back(t.coef);
      }
      r->m[m].rowptr.push_back((uint32_t)r->m[m].col.size());
    }
  }
  LC mul(const LC& a, const LC& b) {
    LC o = var(a.v * b.v);
    enforce(a, b, o);
    return o;
  }
  void enforce_equal(const LC& a, const LC& b) { enforce(sub(a, b), constant(Fr::one()), LC()); }

  // GateInstructions::is_zero: out = 1 iff x == 0   (x * inv = 1 - out, x * out = 0)
  LC is_zero(const LC& x) {
    const bool zero = x.v.is_zero();
    LC inv = var(zero ? Fr::zero() : x.v.inv());
    LC out = var(zero ? Fr::one() : Fr::zero());
    enforce(x, inv, sub(constant(Fr::one()), out));
    enforce(x, out, LC());
    return out;
  }
  // GateInstructions::select(a, b, sel) = sel ? a : b
  LC select(const LC& a, const LC& b, const LC& sel) {
    LC t = mul(sel, sub(a, b));
    return add(t, b);
  }
  // value < 2^bits: bit variables b_i (b_i * (b_i - 1) = 0) with sum b_i 2^i = x
  bool range(const LC& x, int bits) {
    uint8_t raw[32];
    fr_to_wire(x.v, raw);
    bool fits = true;
    for (int i = bits; i < 256; i++)
      if ((raw[i >> 3] >> (i & 7)) & 1) fits = false;
    LC sum;
    Fr pw = Fr::one();
    for (int i = 0; i < bits; i++) {
      LC b = var(((raw[i >> 3] >> (i & 7)) & 1) ? Fr::one() : Fr::zero());
      enforce(b, sub(b, constant(Fr::one())), LC());
      add_scaled(sum, b, pw);
      pw = pw.dbl();
    }
    compact(sum);
    enforce_equal(sum, x);
    return fits;
  }

  LC pow5(const LC& x) {
    LC x2 = mul(x, x);
    LC x4 = mul(x2, x2);
    return mul(x4, x);
  }
  void permute(LC st[POS_T]) {
    const PoseidonFr& p = pos_fr();
    for (int rd = 0; rd < POS_ROUNDS; rd++) {
      const bool full = rd < POS_RF / 2 || rd >= POS_RF / 2 + POS_RP;
      for (int i = 0; i < POS_T; i++) st[i] = add(st[i], constant(p.rc[POS_T * rd + i]));
      st[0] = pow5(st[0]);
      if (full)
        for (int i = 1; i < POS_T; i++) st[i] = pow5(st[i]);
      LC nx[POS_T];
      for (int i = 0; i < POS_T; i++) {
        for (int j = 0; j < POS_T; j++) add_scaled(nx[i], st[j], p.mds[POS_T * i + j]);
        compact(nx[i]);
      }
      for (int i = 0; i < POS_T; i++) st[i] = std::move(nx[i]);
    }
  }
  // PoseidonHasher::hash_fix_len_array
  LC hash(const std::vector<LC>& in) {
    LC st[POS_T];
    st[0] = constant(pos_fr().cap);
    size_t done = 0;
    for (bool more = true; more;) {
      const size_t take = std::min<size_t>(POS_RATE, in.size() - done);
      for (size_t i = 0; i < take; i++) st[1 + i] = add(st[1 + i], in[done + i]);
      if (take < (size_t)POS_RATE) st[1 + take] = add(st[1 + take], constant(Fr::one()));
      done += take;
      permute(st);
      more = take == (size_t)POS_RATE;
    }
    return st[1];
  }
};

// Builds constraints (when r != null) and the assignment for one instance.  Returns the status a
// prover-side caller would see: the mock's ZkpError variants for an impossible update.
struct ChainShape {
  uint64_t K = 0;       // chain variables
  uint32_t n_free = 0;  // unconstrained zero variables behind the chain
};

// Padding shared by the relations: a multiplication chain s_{k+1} = s_k^2 + s_{k-1} seeded from the loaded
// inputs, then re-check rows / free variables, so that constraints + publics = N and
// variables = N exactly.  Returns ZKMI_OK with *shape_out filled when only the shape is asked for.
int32_t pad_chain(Builder& b, zkmi_r1cs* r, uint32_t N, uint32_t n_pub, LC s_prev, LC s_cur, ChainShape* shape_out,
                  bool* shape_only) {
  *shape_only = false;
  const uint32_t N_PUB = n_pub;
  const uint32_t c_real = b.n_constraints, v_real = (uint32_t)b.z.size();
  if ((uint64_t)c_real + N_PUB + 8 > N || (uint64_t)v_real + 8 > N) return ZKMI_ERR_BAD_ARG;  // log_n too small
  // K chain variables, n_re re-check rows (no new variable), n_free unconstrained variables
  int64_t K = (int64_t)N - v_real;
  int64_t n_re = (int64_t)N - N_PUB - c_real - K;
  uint32_t n_free = 0;
  if (n_re < 0) {
    n_free = (uint32_t)(-n_re);
    K -= n_free;
    n_re = 0;
  }
  if (shape_out) {
    shape_out->K = (uint64_t)K;
    shape_out->n_free = n_free;
    *shape_only = true;
    return ZKMI_OK;  // shape query: the relation proper has been walked, the padding is not needed
  }
  if (!r) {
    // assignment only: the chain is a tight value loop (1 squaring + 1 addition per variable)
    b.z.reserve(N);
    Fr sp = s_prev.v, sc = s_cur.v;
    for (int64_t k = 0; k < K; k++) {
      const Fr nx = sc.sqr() + sp;
      b.z.push_back(nx);
      sp = sc;
      sc = nx;
    }
    b.n_constraints += (uint32_t)(K + n_re);
  } else {
    const LC s_m1 = s_prev, s_0 = s_cur;
    std::vector<uint32_t> chain_cols;
    chain_cols.reserve((size_t)K);
    for (int64_t k = 0; k < K; k++) {
      // s_{k+1} = s_k^2 + s_{k-1}   <=>   s_k * s_k = s_{k+1} - s_{k-1}
      LC nxt = b.var(s_cur.v.sqr() + s_prev.v);
      if (r) b.enforce(s_cur, s_cur, Builder::sub(nxt, s_prev));
      else b.n_constraints++;
      chain_cols.push_back(nxt.t[0].col);
      s_prev = std::move(s_cur);
      s_cur = std::move(nxt);
    }
    // re-check rows: (2 s_k) * s_k = 2 s_{k+1} - 2 s_{k-1} over the first n_re chain steps
    {
      auto s_at = [&](int64_t k) -> LC {  // k in [-1, K]
        if (k == -1) return s_m1;
        if (k == 0) return s_0;
        LC l;
        l.t.push_back({chain_cols[(size_t)k - 1], Fr::one()});
        l.v = b.z[chain_cols[(size_t)k - 1]];
        return l;
      };
      const Fr two = fr_small(2);
      for (int64_t k = 0; k < n_re; k++) {
        const int64_t kk = K > 0 ? k % K : 0;
        LC a, c;
        Builder::add_scaled(a, s_at(kk), two);
        Builder::compact(a);
        if (K > 0) {
          Builder::add_scaled(c, s_at(kk + 1), two);
          Builder::add_scaled(c, s_at(kk - 1), two.neg());
          Builder::compact(c);
          b.enforce(a, s_at(kk), c);
        } else {
          b.enforce(LC(), LC(), LC());
        }
      }
    }
  }
  for (uint32_t i = 0; i < n_free; i++) b.var(Fr::zero());

  if (b.z.size() != N || b.n_constraints != N - N_PUB) return ZKMI_ERR_BAD_ARG;
  return ZKMI_OK;
}

int tree_height_of(const zkmi_note_update& in) { return in.tree_height ? (int)in.tree_height : ZKMI_MERKLE_TREE_DEPTH; }

int32_t synthesize(uint32_t log_n, int32_t op_kind, const zkmi_note_update& in, zkmi_r1cs* r, std::vector<Fr>* z_out,
                   ChainShape* shape_out = nullptr) {
  const uint32_t N = 1u << log_n;
  const int TREE_HEIGHT = tree_height_of(in);
  if (TREE_HEIGHT < 1 || TREE_HEIGHT > ZKMI_MAX_TREE_HEIGHT) return ZKMI_ERR_BAD_ARG;
  Builder b(r);
  auto load = [&](const zkmi_fr& f, bool* ok) {
    Fr v;
    if (!fr_from_wire(f.bytes, &v)) *ok = false;
    return v;
  };
  bool ok = true;
  // ---- UpdateNoteInput::new: publics first, then the witnesses in its load order ----
  LC amount = b.var(load(in.amount, &ok)), token = b.var(load(in.token, &ok)), user = b.var(load(in.user, &ok));
  LC new_note_hash = b.var(Fr::zero());  // value patched below (computed from the loaded fields)
  LC merkle_root = b.var(Fr::zero());
  LC old_nullifier = b.var(load(in.old_note[2], &ok));
  LC new_id = b.var(load(in.new_note[0], &ok)), new_trap = b.var(load(in.new_note[1], &ok)),
     new_null = b.var(load(in.new_note[2], &ok));
  LC new_acc_hash = b.var(Fr::zero());
  LC old_id = b.var(load(in.old_note[0], &ok)), old_trap = b.var(load(in.old_note[1], &ok));
  LC old_acc_hash = b.var(Fr::zero());
  LC shape[ZKMI_MAX_TREE_HEIGHT], path[ZKMI_MAX_TREE_HEIGHT];
  for (int i = 0; i < TREE_HEIGHT; i++) {
    if (in.path_shape[i] > 1) ok = false;
    shape[i] = b.var(fr_small(in.path_shape[i]));
  }
  for (int i = 0; i < TREE_HEIGHT; i++) path[i] = b.var(load(in.path[i], &ok));
  LC priv_user = b.var(load(in.op_priv_user, &ok));
  LC acc[4];
  for (int i = 0; i < 4; i++) acc[i] = b.var(load(in.account[i], &ok));
  if (!ok) return ZKMI_ERR_NON_CANONICAL;
  const uint32_t n_loaded = (uint32_t)b.z.size();

  int32_t status = ZKMI_OK;
  // ---- Account::update on the host (mocked_zk/src/account.rs:37-82) -> derived field values ----
  Fr new_bal[2] = {acc[1].v, acc[3].v};
  {
    int hit = -1;
    for (int s = 0; s < 2; s++)
      if (acc[2 * s].v == token.v && hit < 0) hit = s;
    if (hit < 0) status = ZKMI_ERR_ACCOUNT_UPDATE;
    else new_bal[hit] = op_kind == ZKMI_OP_DEPOSIT ? acc[2 * hit + 1].v + amount.v : acc[2 * hit + 1].v - amount.v;
    if (acc[0].v == acc[2].v) status = ZKMI_ERR_ACCOUNT_UPDATE;  // slots must hold distinct tokens
  }
  if (!(priv_user.v == user.v) && status == ZKMI_OK) status = ZKMI_ERR_OPERATION_COMBINE;

  // ---- update_note_circuit ----
  // patch the derived note fields before they are hashed: the circuit recomputes them
  auto set_value = [&](LC& l, const Fr& v) {
    l.v = v;
    b.z[l.t[0].col] = v;
  };
  {
    // a throw-away builder computes the hashes without touching the constraint system
    Builder h(nullptr);
    LC oa = h.hash({Builder::constant(acc[0].v), Builder::constant(acc[1].v), Builder::constant(acc[2].v),
                    Builder::constant(acc[3].v)});
    LC na = h.hash({Builder::constant(acc[0].v), Builder::constant(new_bal[0]), Builder::constant(acc[2].v),
                    Builder::constant(new_bal[1])});
    set_value(old_acc_hash, oa.v);
    set_value(new_acc_hash, na.v);
    LC nn = h.hash({Builder::constant(new_id.v), Builder::constant(new_trap.v), Builder::constant(new_null.v),
                    Builder::constant(na.v)});
    set_value(new_note_hash, nn.v);
    LC cur = h.hash({Builder::constant(old_id.v), Builder::constant(old_trap.v), Builder::constant(old_nullifier.v),
                     Builder::constant(oa.v)});
    for (int i = 0; i < TREE_HEIGHT; i++) {
      const bool sel = shape[i].v.is_zero();
      LC left = Builder::constant(sel ? path[i].v : cur.v), right = Builder::constant(sel ? cur.v : path[i].v);
      cur = h.hash({left, right});
    }
    set_value(merkle_root, cur.v);
  }

  // verify_note_circuit(new_note, new_note_hash)
  b.enforce_equal(b.hash({new_id, new_trap, new_null, new_acc_hash}), new_note_hash);
  // old_note_hash, Merkle proof up to merkle_root
  LC cur = b.hash({old_id, old_trap, old_nullifier, old_acc_hash});
  for (int i = 0; i < TREE_HEIGHT; i++) {
    LC sel = b.is_zero(shape[i]);
    LC left = b.select(path[i], cur, sel);
    LC right = b.select(cur, path[i], sel);
    cur = b.hash({left, right});
  }
  b.enforce_equal(cur, merkle_root);
  // CircuitOperation::combine(op_priv, op_pub): same user
  b.enforce_equal(priv_user, user);
  // update_account_circuit: verify old account, update, verify new account
  b.enforce_equal(b.hash({acc[0], acc[1], acc[2], acc[3]}), old_acc_hash);
  LC m0 = b.is_zero(Builder::sub(acc[0], token)), m1 = b.is_zero(Builder::sub(acc[2], token));
  b.enforce_equal(Builder::add(m0, m1), Builder::constant(Fr::one()));
  LC d0 = b.mul(m0, amount), d1 = b.mul(m1, amount);
  LC nb0 = op_kind == ZKMI_OP_DEPOSIT ? Builder::add(acc[1], d0) : Builder::sub(acc[1], d0);
  LC nb1 = op_kind == ZKMI_OP_DEPOSIT ? Builder::add(acc[3], d1) : Builder::sub(acc[3], d1);
  // checked_add / checked_sub on u128 (account.rs:46-49, 66-69)
  const bool fit0 = b.range(nb0, BALANCE_BITS), fit1 = b.range(nb1, BALANCE_BITS);
  if (!(fit0 && fit1) && status == ZKMI_OK) status = ZKMI_ERR_ACCOUNT_UPDATE;
  b.enforce_equal(b.hash({acc[0], nb0, acc[2], nb1}), new_acc_hash);

  // ---- padding chain up to constraints + publics = N, variables = N ----
  LC s_prev, s_cur;
  {
    const LC* pub[4] = {&amount, &token, &user, &old_nullifier};
    for (int j = 0; j < 4; j++) Builder::add_scaled(s_prev, *pub[j], fr_small(j + 1));
    Builder::compact(s_prev);
    for (uint32_t col = N_PUB; col < n_loaded; col++) {
      LC one_var;
      one_var.t.push_back({col, Fr::one()});
      one_var.v = b.z[col];
      Builder::add_scaled(s_cur, one_var, fr_small(col - N_PUB + 1));
    }
    Builder::compact(s_cur);
  }
  bool shape_only = false;
  const int32_t prc = pad_chain(b, r, N, N_PUB, std::move(s_prev), std::move(s_cur), shape_out, &shape_only);
  if (prc != ZKMI_OK || shape_only) return prc;
  if (r) {
    r->n_vars = N;
    r->n_pub = N_PUB;
    r->n_constraints = b.n_constraints;
    r1cs_finish_shape(r);
  }
  if (z_out) *z_out = std::move(b.z);
  return status;
}

// ---- the creation relation: what ZkProof::verify_creation stands for -------------------------------
// mocked_zk/src/relations.rs:127-136 (consumed at contract/lib.rs:50-58): the new note commits to
// (id, trapdoor, nullifier, hash(Account::new(tokens))).  Circuit pieces: verify_account_circuit
// (update_account.rs:52-65) on the fresh account [(token_0, 0), (token_1, 0)] (account.rs:27-34) and
// verify_note_circuit (update_note.rs:91-103).  Publics: h_note_new || tokens (the arguments of
// verify_creation, in its order); witnesses: id, trapdoor, nullifier, account hash.
constexpr uint32_t N_PUB_CREATE = 4;  // 1, h_note_new, token_0, token_1

int32_t synthesize_create(uint32_t log_n, const zkmi_note_create& in, zkmi_r1cs* r, std::vector<Fr>* z_out) {
  const uint32_t N = 1u << log_n;
  Builder b(r);
  bool ok = true;
  auto load = [&](const zkmi_fr& f) {
    Fr v;
    if (!fr_from_wire(f.bytes, &v)) ok = false;
    return v;
  };
  LC h_note_new = b.var(Fr::zero());  // patched below
  LC tok0 = b.var(load(in.tokens[0])), tok1 = b.var(load(in.tokens[1]));
  LC id = b.var(load(in.note[0])), trap = b.var(load(in.note[1])), null = b.var(load(in.note[2]));
  LC acc_hash = b.var(Fr::zero());
  if (!ok) return ZKMI_ERR_NON_CANONICAL;
  const uint32_t n_loaded = (uint32_t)b.z.size();
  {
    Builder h(nullptr);
    const LC zero = Builder::constant(Fr::zero());
    LC ah = h.hash({Builder::constant(tok0.v), zero, Builder::constant(tok1.v), zero});
    acc_hash.v = ah.v;
    b.z[acc_hash.t[0].col] = ah.v;
    LC nh = h.hash({Builder::constant(id.v), Builder::constant(trap.v), Builder::constant(null.v), Builder::constant(ah.v)});
    h_note_new.v = nh.v;
    b.z[h_note_new.t[0].col] = nh.v;
  }
  // verify_account_circuit(Account::new(tokens), account_hash): balances are the constant 0
  b.enforce_equal(b.hash({tok0, Builder::constant(Fr::zero()), tok1, Builder::constant(Fr::zero())}), acc_hash);
  // verify_note_circuit(new_note, h_note_new)
  b.enforce_equal(b.hash({id, trap, null, acc_hash}), h_note_new);

  LC s_prev, s_cur;
  Builder::add_scaled(s_prev, tok0, fr_small(1));
  Builder::add_scaled(s_prev, tok1, fr_small(2));
  Builder::compact(s_prev);
  for (uint32_t col = N_PUB_CREATE; col < n_loaded; col++) {
    LC one_var;
    one_var.t.push_back({col, Fr::one()});
    one_var.v = b.z[col];
    Builder::add_scaled(s_cur, one_var, fr_small(col - N_PUB_CREATE + 1));
  }
  Builder::compact(s_cur);
  bool shape_only = false;
  const int32_t prc = pad_chain(b, r, N, N_PUB_CREATE, std::move(s_prev), std::move(s_cur), nullptr, &shape_only);
  if (prc != ZKMI_OK) return prc;
  if (r) {
    r->n_vars = N;
    r->n_pub = N_PUB_CREATE;
    r->n_constraints = b.n_constraints;
    r1cs_finish_shape(r);
  }
  if (z_out) *z_out = std::move(b.z);
  return ZKMI_OK;
}

// mock Scalar (any 32 bytes) -> Fr element: the value mod r
zkmi_fr fr_of_scalar(const zkmi_scalar& s) {
  zkmi_fr o;
  zkmi_fr_reduce(s.bytes, o.bytes);
  return o;
}
zkmi_fr fr_of_u128(const uint8_t le16[16]) {
  zkmi_fr o;
  memset(o.bytes, 0, 32);
  memcpy(o.bytes, le16, 16);
  return o;
}

// K / n_free of the padding for (log_n, op_kind, tree height): walk the relation proper once on a fixed instance
int32_t chain_shape(uint32_t log_n, int32_t op_kind, int height, ChainShape* out) {
  // K = 2^log_n - (variables of the relation proper), the same walk for every log_n: cache per (kind, height)
  static std::mutex mu;
  static bool known[2][ZKMI_MAX_TREE_HEIGHT + 1] = {};
  static uint64_t v_real[2][ZKMI_MAX_TREE_HEIGHT + 1];
  static uint32_t n_free[2][ZKMI_MAX_TREE_HEIGHT + 1];
  if (height < 1 || height > ZKMI_MAX_TREE_HEIGHT) return ZKMI_ERR_BAD_ARG;
  {
    std::lock_guard<std::mutex> g(mu);
    if (known[op_kind][height]) {
      if ((1ull << log_n) < v_real[op_kind][height] + 16) return ZKMI_ERR_BAD_ARG;  // log_n too small
      out->n_free = n_free[op_kind][height];
      out->K = (1ull << log_n) - v_real[op_kind][height] - n_free[op_kind][height];
      return ZKMI_OK;
    }
  }
  zkmi_note_update in;
  memset(&in, 0, sizeof(in));
  in.account[2].bytes[0] = 1;
  in.tree_height = (uint32_t)height;
  const int32_t rc = synthesize(log_n, op_kind, in, nullptr, nullptr, out);
  if (rc == ZKMI_OK) {
    std::lock_guard<std::mutex> g(mu);
    n_free[op_kind][height] = out->n_free;
    v_real[op_kind][height] = (1ull << log_n) - out->K - out->n_free;
    known[op_kind][height] = true;
  }
  return rc;
}

// one thread = one instance (the statement sequence is sequential; a batch supplies the parallelism)
__global__ __launch_bounds__(64) void k_update_note_values(const zkmi_note_update* __restrict__ in, uint32_t n,
                                                          int32_t op_kind, int height, uint64_t K, uint32_t n_free,
                                                          const PoseidonConsts<Fr28>* __restrict__ c,
                                                          uint32_t* const* __restrict__ z_out,
                                                          int32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  status[i] = rv_update_note(in[i], op_kind, height, K, n_free, c, z_out[i]);
}

}  // namespace
}  // namespace zkmi

using namespace zkmi;

extern "C" {

int32_t zkmi_update_note_r1cs_h(uint32_t log_n, int32_t op_kind, uint32_t tree_height, zkmi_r1cs** out) {
  if (!out || log_n < 13 || log_n > 26 || (op_kind != ZKMI_OP_DEPOSIT && op_kind != ZKMI_OP_WITHDRAW) ||
      tree_height < 1 || tree_height > ZKMI_MAX_TREE_HEIGHT)
    return ZKMI_ERR_BAD_ARG;
  *out = nullptr;
  zkmi_r1cs* r = new (std::nothrow) zkmi_r1cs();
  if (!r) return ZKMI_ERR_BAD_ARG;
  // the shape does not depend on the values: synthesize over a fixed valid instance
  zkmi_note_update in;
  memset(&in, 0, sizeof(in));
  in.account[2].bytes[0] = 1;  // distinct token ids 0 and 1
  in.tree_height = tree_height;
  const int32_t rc = synthesize(log_n, op_kind, in, r, nullptr);
  if (rc != ZKMI_OK) {
    delete r;
    return rc;
  }
  r->tree_height = tree_height;
  *out = r;
  return ZKMI_OK;
}
int32_t zkmi_update_note_r1cs(uint32_t log_n, int32_t op_kind, zkmi_r1cs** out) {
  return zkmi_update_note_r1cs_h(log_n, op_kind, ZKMI_MERKLE_TREE_DEPTH, out);
}

int32_t zkmi_update_note_witness(uint32_t log_n, int32_t op_kind, const zkmi_note_update* in, uint8_t* out_z,
                                 uint8_t* out_publics) {
  if (!in || log_n < 13 || log_n > 26 || (op_kind != ZKMI_OP_DEPOSIT && op_kind != ZKMI_OP_WITHDRAW))
    return ZKMI_ERR_BAD_ARG;
  std::vector<Fr> z;
  const int32_t rc = synthesize(log_n, op_kind, *in, nullptr, &z);
  if (rc == ZKMI_ERR_BAD_ARG || rc == ZKMI_ERR_NON_CANONICAL) return rc;
  if (out_z)
    for (size_t i = 0; i < z.size(); i++) fr_to_wire(z[i], out_z + 32 * i);
  if (out_publics)
    for (uint32_t i = 1; i < N_PUB; i++) fr_to_wire(z[i], out_publics + 32 * (i - 1));
  return rc;  // ZKMI_OK, or the mock's error for an update the relation cannot satisfy
}

// The device code path (relation_values.hpp) executed on the host for ONE instance: used by the
// CPU tests to pin it against the constraint builder above.
int32_t zkmi_update_note_witness_values_host(uint32_t log_n, int32_t op_kind, const zkmi_note_update* in, uint8_t* out_z) {
  if (!in || !out_z || log_n < 13 || log_n > 26 || (op_kind != ZKMI_OP_DEPOSIT && op_kind != ZKMI_OP_WITHDRAW))
    return ZKMI_ERR_BAD_ARG;
  ChainShape sh;
  const int height = tree_height_of(*in);
  int32_t rc = chain_shape(log_n, op_kind, height, &sh);
  if (rc != ZKMI_OK) return rc;
  std::vector<uint32_t> z((size_t)8 << log_n);
  rc = rv_update_note(*in, op_kind, height, sh.K, sh.n_free, poseidon_consts_bls(), z.data());
  memcpy(out_z, z.data(), (size_t)32 << log_n);
  return rc;
}

// Batch of n instances on the device: inputs are host structs, d_z_out[i] are device buffers of
// 2^log_n x 32 B; out_status[i] receives ZKMI_OK or the mock's error code per instance.
int32_t zkmi_update_note_witness_batch_dev(zkmi_ctx* ctx, uint32_t log_n, int32_t op_kind, const zkmi_note_update* in,
                                           uint32_t n, void* const* d_z_out, int32_t* out_status) {
  ZK_ENTER(ctx);
  if (log_n < 13 || log_n > 26 || (op_kind != ZKMI_OP_DEPOSIT && op_kind != ZKMI_OP_WITHDRAW) ||
      (n && (!in || !d_z_out || !out_status)))
    return ZKMI_ERR_BAD_ARG;
  if (n == 0) return ZKMI_OK;
  // one launch = one relation shape: every instance of the batch must name the same tree height
  const int height = tree_height_of(in[0]);
  for (uint32_t i = 1; i < n; i++)
    if (tree_height_of(in[i]) != height) return ctx->fail(ZKMI_ERR_BAD_ARG, "mixed tree heights in one batch");
  ChainShape sh;
  const int32_t rc = chain_shape(log_n, op_kind, height, &sh);
  if (rc != ZKMI_OK) return rc;
  if (!ctx->d_pos[ZKMI_FIELD_BLS12_381_FR]) {
    ZK_HIP(ctx, hipMalloc(&ctx->d_pos[ZKMI_FIELD_BLS12_381_FR], sizeof(PoseidonConsts<Fr28>)));
    ZK_HIP(ctx, hipMemcpy(ctx->d_pos[ZKMI_FIELD_BLS12_381_FR], poseidon_consts_bls(), sizeof(PoseidonConsts<Fr28>),
                          hipMemcpyHostToDevice));
  }
  const size_t in_bytes = sizeof(zkmi_note_update) * (size_t)n, ptr_bytes = sizeof(void*) * (size_t)n;
  const size_t off_ptr = (in_bytes + 255) & ~(size_t)255, off_st = (off_ptr + ptr_bytes + 255) & ~(size_t)255;
  ZK_HIP(ctx, ctx->staging(off_st + sizeof(int32_t) * (size_t)n));
  uint8_t* d = static_cast<uint8_t*>(ctx->d_tmp);
  ZK_HIP(ctx, hipMemcpyAsync(d, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(ctx, hipMemcpyAsync(d + off_ptr, d_z_out, ptr_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->timer()) ctx->timer()->begin(PH_WITNESS, ctx->stream);
  hipLaunchKernelGGL(k_update_note_values, dim3((n + 63) / 64), dim3(64), 0, ctx->stream,
                     reinterpret_cast<const zkmi_note_update*>(d), n, op_kind, height, sh.K, sh.n_free,
                     static_cast<const PoseidonConsts<Fr28>*>(ctx->d_pos[ZKMI_FIELD_BLS12_381_FR]),
                     reinterpret_cast<uint32_t* const*>(d + off_ptr), reinterpret_cast<int32_t*>(d + off_st));
  if (ctx->timer()) ctx->timer()->end(PH_WITNESS, ctx->stream);
  ZK_HIP(ctx, hipGetLastError());
  ZK_HIP(ctx, hipMemcpyAsync(out_status, d + off_st, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// ---- creation relation + the ZkProof surface with real proofs -----------------------
int32_t zkmi_create_note_r1cs(uint32_t log_n, zkmi_r1cs** out) {
  if (!out || log_n < 11 || log_n > 26) return ZKMI_ERR_BAD_ARG;
  *out = nullptr;
  zkmi_r1cs* r = new (std::nothrow) zkmi_r1cs();
  if (!r) return ZKMI_ERR_BAD_ARG;
  zkmi_note_create in;
  memset(&in, 0, sizeof(in));
  const int32_t rc = synthesize_create(log_n, in, r, nullptr);
  if (rc != ZKMI_OK) {
    delete r;
    return rc;
  }
  *out = r;
  return ZKMI_OK;
}

int32_t zkmi_create_note_witness(uint32_t log_n, const zkmi_note_create* in, uint8_t* out_z, uint8_t* out_publics) {
  if (!in || log_n < 11 || log_n > 26) return ZKMI_ERR_BAD_ARG;
  std::vector<Fr> z;
  const int32_t rc = synthesize_create(log_n, *in, nullptr, &z);
  if (rc != ZKMI_OK) return rc;
  if (out_z)
    for (size_t i = 0; i < z.size(); i++) fr_to_wire(z[i], out_z + 32 * i);
  if (out_publics)
    for (uint32_t i = 1; i < N_PUB_CREATE; i++) fr_to_wire(z[i], out_publics + 32 * (i - 1));
  return ZKMI_OK;
}

// ZkProof::new + verify_creation with a real proof (relations.rs:37-55, :127-136; callers
// drink_tests/utils/shielder.rs:60 and contract/lib.rs:56)
int32_t zkmi_shielder_prove_creation(zkmi_ctx* ctx, const zkmi_pk* pk_create, const zkmi_zkproof* knowledge,
                                     const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], const uint8_t r[32], const uint8_t s[32],
                                     zkmi_scalar* out_h_note_new, uint8_t out_proof[192]) {
  ZK_ENTER(ctx);
  if (!pk_create || !knowledge || !tokens || !r || !s || !out_h_note_new || !out_proof) return ZKMI_ERR_BAD_ARG;
  uint32_t n_vars = 0, n_pub = 0, log_n = 0;
  int32_t rc = zkmi_pk_shape(pk_create, &n_vars, &n_pub, &log_n);
  if (rc != ZKMI_OK || n_pub != N_PUB_CREATE) return ctx->fail(ZKMI_ERR_BAD_ARG, "not a creation-relation key");
  zkmi_note_create in;
  in.tokens[0] = fr_of_scalar(tokens[0]);
  in.tokens[1] = fr_of_scalar(tokens[1]);
  in.note[0] = fr_of_scalar(knowledge->id);
  in.note[1] = fr_of_scalar(knowledge->trapdoor_new);
  in.note[2] = fr_of_scalar(knowledge->nullifier_new);
  std::vector<uint8_t> z((size_t)32 << log_n);
  uint8_t pub[96];
  if ((rc = zkmi_create_note_witness(log_n, &in, z.data(), pub)) != ZKMI_OK) return rc;
  if ((rc = zkmi_groth16_prove(ctx, pk_create, z.data(), r, s, out_proof)) != ZKMI_OK) return rc;
  memcpy(out_h_note_new->bytes, pub, 32);
  return ZKMI_OK;
}

int32_t zkmi_shielder_verify_creation(const uint8_t* vk_create, const zkmi_scalar* h_note_new,
                                      const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], const uint8_t proof[192]) {
  if (!vk_create || !h_note_new || !tokens || !proof) return ZKMI_ERR_BAD_ARG;
  uint8_t pub[96];
  memcpy(pub, h_note_new->bytes, 32);  // a hash output: must already be canonical, the verifier checks
  memcpy(pub + 32, fr_of_scalar(tokens[0]).bytes, 32);
  memcpy(pub + 64, fr_of_scalar(tokens[1]).bytes, 32);
  const int32_t rc = zkmi_groth16_verify(vk_create, N_PUB_CREATE, pub, proof);
  return rc == ZKMI_OK ? ZKMI_OK : ZKMI_ERR_VERIFICATION;  // ZkpError::VerificationError, like verify_hash
}

// ZkProof::update_account with a real proof (relations.rs:79-98; caller drink_tests/utils/shielder.rs:105-114).
// The state transition is the mock's (account update, transition()); the note hashes and the Merkle
// root are the relation's Poseidon values; OpPub -> publics as ops.rs:6-25 + update_note.rs:121,127.
int32_t zkmi_shielder_prove_update(zkmi_ctx* ctx, const zkmi_pk* pk_deposit, const zkmi_pk* pk_withdraw,
                                   const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv,
                                   const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                                   const zkmi_scalar* merkle_proof, uint32_t tree_height, uint32_t leaf_id,
                                   const uint8_t r[32], const uint8_t s[32], zkmi_scalar* out_h_note_new,
                                   zkmi_scalar* out_merkle_root, zkmi_zkproof* out_new, uint8_t out_proof[192]) {
  ZK_ENTER(ctx);
  if (!self || !op_pub || !op_priv || !trapdoor || !nullifier || !merkle_proof || !r || !s || !out_h_note_new ||
      !out_proof || op_pub->kind > 1 || tree_height < 1 || tree_height > ZKMI_MAX_TREE_HEIGHT)
    return ZKMI_ERR_BAD_ARG;
  const zkmi_pk* pk = op_pub->kind == 0 ? pk_deposit : pk_withdraw;
  if (!pk) return ctx->fail(ZKMI_ERR_BAD_ARG, "no proving key for this operation kind");
  // Operation::combine, Account::update: the mock's checks and error codes, before any GPU work
  int32_t rc = zkmi_operation_combine(op_pub, op_priv);
  if (rc != ZKMI_OK) return rc;
  zkmi_account acc_updated;
  if ((rc = zkmi_account_update(&self->acc_new, op_pub, op_priv, &acc_updated)) != ZKMI_OK) return rc;
  uint32_t n_vars = 0, n_pub = 0, log_n = 0;
  if ((rc = zkmi_pk_shape(pk, &n_vars, &n_pub, &log_n)) != ZKMI_OK || n_pub != N_PUB)
    return ctx->fail(ZKMI_ERR_BAD_ARG, "not an update_note key");
  // a key made for another Merkle height can only end in ZKMI_ERR_UNSATISFIED after a full GPU proof: say so up front
  if (pk_tree_height(pk) && pk_tree_height(pk) != tree_height)
    return ctx->fail(ZKMI_ERR_BAD_ARG, "tree_height differs from the height the proving key's relation was built for");
  // the returned ZkProof keeps the mock's fixed-depth path (relations.rs:25: [Scalar; MERKLE_TREE_DEPTH]); a deeper
  // path does not fit it, and a truncated one could never reproduce the root
  if (out_new && tree_height > ZKMI_MERKLE_TREE_DEPTH)
    return ctx->fail(ZKMI_ERR_BAD_ARG, "out_new holds a depth-10 path: pass NULL for deeper trees");
  zkmi_note_update in;
  memset(&in, 0, sizeof(in));
  in.amount = fr_of_u128(op_pub->amount);
  in.token = fr_of_scalar(op_pub->token);
  in.user = fr_of_scalar(op_pub->user);
  in.new_note[0] = fr_of_scalar(self->id);
  in.new_note[1] = fr_of_scalar(*trapdoor);
  in.new_note[2] = fr_of_scalar(*nullifier);
  in.old_note[0] = fr_of_scalar(self->id);
  in.old_note[1] = fr_of_scalar(self->trapdoor_new);   // transition(): the current note becomes the old one
  in.old_note[2] = fr_of_scalar(self->nullifier_new);
  in.tree_height = tree_height;
  for (uint32_t i = 0; i < tree_height; i++) {
    // verify_merkle_proof (relations.rs:110-125): even index -> the running node is the left input;
    // path_shape = 0 means "sibling on the left" (merkle_proof.rs:53-55)
    in.path_shape[i] = (uint8_t)(1u - ((leaf_id >> i) & 1u));
    in.path[i] = fr_of_scalar(merkle_proof[i]);
  }
  in.op_priv_user = fr_of_scalar(op_priv->user);
  for (int t = 0; t < ZKMI_TOKENS_NUMBER; t++) {
    in.account[2 * t] = fr_of_scalar(self->acc_new.balances[t][0]);
    in.account[2 * t + 1] = fr_of_scalar(self->acc_new.balances[t][1]);
  }
  std::vector<uint8_t> z((size_t)32 << log_n);
  uint8_t pub[192];
  if ((rc = zkmi_update_note_witness(log_n, (int32_t)op_pub->kind, &in, z.data(), pub)) != ZKMI_OK) return rc;
  if ((rc = zkmi_groth16_prove(ctx, pk, z.data(), r, s, out_proof)) != ZKMI_OK) return rc;
  memcpy(out_h_note_new->bytes, pub + 96, 32);
  if (out_merkle_root) memcpy(out_merkle_root->bytes, pub + 128, 32);
  if (out_new) {
    // transition() (relations.rs:57-77)
    *out_new = *self;
    out_new->trapdoor_old = self->trapdoor_new;
    out_new->trapdoor_new = *trapdoor;
    out_new->nullifier_new = *nullifier;
    out_new->acc_old = self->acc_new;
    out_new->acc_new = acc_updated;
    out_new->op_priv = *op_priv;
    memset(out_new->merkle_proof, 0, sizeof(out_new->merkle_proof));
    for (uint32_t i = 0; i < tree_height; i++) out_new->merkle_proof[i] = merkle_proof[i];  // tree_height <= depth: checked above
    out_new->merkle_proof_leaf_id = leaf_id;
  }
  return ZKMI_OK;
}

// verify_update with a real proof (relations.rs:138-155; caller contract/lib.rs:74)
int32_t zkmi_shielder_verify_update(const uint8_t* vk_deposit, const uint8_t* vk_withdraw, const zkmi_op_pub* op_pub,
                                    const zkmi_scalar* h_note_new, const zkmi_scalar* merkle_root,
                                    const zkmi_scalar* nullifier_old, const uint8_t proof[192]) {
  if (!op_pub || !h_note_new || !merkle_root || !nullifier_old || !proof || op_pub->kind > 1) return ZKMI_ERR_BAD_ARG;
  const uint8_t* vk = op_pub->kind == 0 ? vk_deposit : vk_withdraw;
  if (!vk) return ZKMI_ERR_BAD_ARG;
  uint8_t pub[192];
  memcpy(pub, fr_of_u128(op_pub->amount).bytes, 32);
  memcpy(pub + 32, fr_of_scalar(op_pub->token).bytes, 32);
  memcpy(pub + 64, fr_of_scalar(op_pub->user).bytes, 32);
  memcpy(pub + 96, h_note_new->bytes, 32);
  memcpy(pub + 128, merkle_root->bytes, 32);
  memcpy(pub + 160, fr_of_scalar(*nullifier_old).bytes, 32);
  const int32_t rc = zkmi_groth16_verify(vk, N_PUB, pub, proof);
  return rc == ZKMI_OK ? ZKMI_OK : ZKMI_ERR_VERIFICATION;  // contract/errors.rs:24-28 flattens every ZkpError anyway
}

}  // extern "C"

// zkmi — value-only synthesis of the update_note relation ("witness generation
// on device").  Walks exactly the statement sequence of relation.hip's Builder — same gadgets, same
// order of variable allocation — but carries plain field values and streams every allocated
// variable to a sink in wire format.  __host__ __device__: one thread produces one instance's
// assignment; the kernel in relation.hip runs it for a batch, and the host execution of the same
// code is compared with the Builder's assignment in the CPU tests.
//
// Reference statements mirrored: update_note.rs:47-88 (load order), :91-103, :106-149,
// merkle_proof.rs:38-61, update_account.rs:68-95, mocked_zk account.rs:37-82 / ops.rs:47-63.
#pragma once
#include "../../include/zkmi.h"
#include "poseidon.hpp"

namespace zkmi {

constexpr int RV_BALANCE_BITS = 128;
constexpr uint32_t RV_N_PUB = 7;
// loaded by UpdateNoteInput::new: 1 + 6 publics + 7 note fields + 2 H path entries + op_priv + 4 account
// fields = 19 + 2 H variables for tree height H (39 at the mock's depth 10)

// streams 32-byte canonical little-endian elements into one assignment vector
struct WireSink {
  uint32_t* base;
  uint64_t idx;
  ZK_HD void set(uint64_t at, const Fr28& v) {
    uint32_t w[8];
    v.to_canonical(w);
    uint32_t* o = base + 8 * at;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = w[k];
  }
  ZK_HD void put(const Fr28& v) { set(idx++, v); }
};
struct NullSink {
  ZK_HD void put(const Fr28&) {}
};

ZK_HD Fr28 rv_small(uint32_t x) {
  uint32_t w[8] = {x, 0, 0, 0, 0, 0, 0, 0};
  return Fr28::from_canonical(w);
}
ZK_HD Fr28 rv_load(const zkmi_fr& f) {
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; k++)
    w[k] = (uint32_t)f.bytes[4 * k] | ((uint32_t)f.bytes[4 * k + 1] << 8) | ((uint32_t)f.bytes[4 * k + 2] << 16) |
           ((uint32_t)f.bytes[4 * k + 3] << 24);
  return Fr28::from_canonical(w);
}
ZK_HD bool rv_canonical(const zkmi_fr& f) {
  for (int k = 7; k >= 0; k--) {
    const uint32_t w = (uint32_t)f.bytes[4 * k] | ((uint32_t)f.bytes[4 * k + 1] << 8) |
                       ((uint32_t)f.bytes[4 * k + 2] << 16) | ((uint32_t)f.bytes[4 * k + 3] << 24);
    if (w != Fr28Params::MOD32[k]) return w < Fr28Params::MOD32[k];
  }
  return false;
}
ZK_HD bool rv_is_zero(const Fr28& v) {
  uint32_t w[8];
  v.to_canonical(w);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc |= w[k];
  return acc == 0;
}

// x^5 with the three product variables the circuit allocates (x2, x4, x5)
template <class SINK>
ZK_HD Fr28 rv_pow5(const Fr28& x, SINK& out) {
  const Fr28 x2 = x.sqr_inline();
  out.put(x2);
  const Fr28 x4 = x2.sqr_inline();
  out.put(x4);
  const Fr28 x5 = Fr28::mul_inline(x4, x);
  out.put(x5);
  return x5;
}

// the plain 64-round permutation, as the circuit lays it out
template <class SINK>
ZK_HD void rv_permute(Fr28 st[POS_T], const PoseidonConsts<Fr28>* __restrict__ c, SINK& out) {
#pragma unroll 1
  for (int rd = 0; rd < POS_ROUNDS; rd++) {
    const bool full = rd < POS_RF / 2 || rd >= POS_RF / 2 + POS_RP;
#pragma unroll
    for (int i = 0; i < POS_T; i++) st[i] = st[i] + c->rc[POS_T * rd + i];
    st[0] = rv_pow5(st[0], out);
    if (full) {
      for (int i = 1; i < POS_T; i++) st[i] = rv_pow5(st[i], out);
    }
    Fr28 nx[POS_T];
    for (int i = 0; i < POS_T; i++) nx[i] = pos_dot5(c->mds + POS_T * i, st);
#pragma unroll
    for (int i = 0; i < POS_T; i++) st[i] = nx[i];
  }
}

// PoseidonHasher::hash_fix_len_array over n <= 4 values (all the relation needs)
template <class SINK>
ZK_HD Fr28 rv_hash(const Fr28* in, int n, const PoseidonConsts<Fr28>* __restrict__ c, SINK& out) {
  Fr28 st[POS_T];
  st[0] = c->cap;
#pragma unroll
  for (int i = 1; i < POS_T; i++) st[i] = Fr28::zero();
  for (int i = 0; i < n; i++) st[1 + i] = st[1 + i] + in[i];
  if (n < POS_RATE) st[1 + n] = st[1 + n] + Fr28::one();
  rv_permute(st, c, out);
  if (n == POS_RATE) {
    st[1] = st[1] + Fr28::one();
    rv_permute(st, c, out);
  }
  return st[1];
}

// GateInstructions::is_zero: variables inv, out
template <class SINK>
ZK_HD Fr28 rv_is_zero(const Fr28& x, SINK& out) {
  const bool zero = rv_is_zero(x);
  out.put(zero ? Fr28::zero() : x.inv());
  const Fr28 o = zero ? Fr28::one() : Fr28::zero();
  out.put(o);
  return o;
}
// GateInstructions::select(a, b, sel): variable t = sel * (a - b)
template <class SINK>
ZK_HD Fr28 rv_select(const Fr28& a, const Fr28& b, const Fr28& sel, SINK& out) {
  const Fr28 t = Fr28::mul_inline(sel, a - b);
  out.put(t);
  return t + b;
}
// 128 bit variables; false if the value does not fit
template <class SINK>
ZK_HD bool rv_range(const Fr28& x, SINK& out) {
  uint32_t w[8];
  x.to_canonical(w);
  const bool fits = (w[4] | w[5] | w[6] | w[7]) == 0;
  const Fr28 one = Fr28::one(), zero = Fr28::zero();
  for (int i = 0; i < RV_BALANCE_BITS; i++) out.put(((w[i >> 5] >> (i & 31)) & 1u) ? one : zero);
  return fits;
}

// One instance: writes n_vars = 39 + gadget variables + K chain variables + n_free zeros, returns
// ZKMI_OK or the mock's error code for an update the relation cannot satisfy.
ZK_HD int32_t rv_update_note(const zkmi_note_update& in, int32_t op_kind, int RV_TREE_HEIGHT, uint64_t K, uint32_t n_free,
                             const PoseidonConsts<Fr28>* __restrict__ c, uint32_t* z_out) {
  bool ok = rv_canonical(in.amount) && rv_canonical(in.token) && rv_canonical(in.user) && rv_canonical(in.op_priv_user);
  for (int i = 0; i < 3; i++) ok = ok && rv_canonical(in.new_note[i]) && rv_canonical(in.old_note[i]);
  for (int i = 0; i < RV_TREE_HEIGHT; i++) ok = ok && rv_canonical(in.path[i]) && in.path_shape[i] <= 1;
  for (int i = 0; i < 4; i++) ok = ok && rv_canonical(in.account[i]);
  if (!ok) return ZKMI_ERR_NON_CANONICAL;

  const Fr28 amount = rv_load(in.amount), token = rv_load(in.token), user = rv_load(in.user);
  const Fr28 old_null = rv_load(in.old_note[2]);
  const Fr28 new_id = rv_load(in.new_note[0]), new_trap = rv_load(in.new_note[1]), new_null = rv_load(in.new_note[2]);
  const Fr28 old_id = rv_load(in.old_note[0]), old_trap = rv_load(in.old_note[1]);
  const Fr28 priv_user = rv_load(in.op_priv_user);
  Fr28 acc[4];
  for (int i = 0; i < 4; i++) acc[i] = rv_load(in.account[i]);

  // Account::update / Operation::combine on plain values
  int32_t status = ZKMI_OK;
  const bool hit0 = rv_is_zero(acc[0] - token), hit1 = rv_is_zero(acc[2] - token);
  Fr28 nb0 = acc[1], nb1 = acc[3];
  if (hit0) nb0 = op_kind == ZKMI_OP_DEPOSIT ? acc[1] + amount : acc[1] - amount;
  else if (hit1) nb1 = op_kind == ZKMI_OP_DEPOSIT ? acc[3] + amount : acc[3] - amount;
  else status = ZKMI_ERR_ACCOUNT_UPDATE;
  if (rv_is_zero(acc[0] - acc[2])) status = ZKMI_ERR_ACCOUNT_UPDATE;
  if (!rv_is_zero(priv_user - user) && status == ZKMI_OK) status = ZKMI_ERR_OPERATION_COMBINE;

  // the two account hashes are loaded witnesses of the notes: compute them first, without emission
  NullSink quiet;
  const Fr28 new_vec[4] = {acc[0], nb0, acc[2], nb1};
  const Fr28 old_acc_hash = rv_hash(acc, 4, c, quiet);
  const Fr28 new_acc_hash = rv_hash(new_vec, 4, c, quiet);

  WireSink out{z_out, 0};
  out.put(Fr28::one());
  out.put(amount);
  out.put(token);
  out.put(user);
  out.put(Fr28::zero());  // new_note_hash, patched below
  out.put(Fr28::zero());  // merkle_root, patched below
  out.put(old_null);
  out.put(new_id);
  out.put(new_trap);
  out.put(new_null);
  out.put(new_acc_hash);
  out.put(old_id);
  out.put(old_trap);
  out.put(old_acc_hash);
  for (int i = 0; i < RV_TREE_HEIGHT; i++) out.put(rv_small(in.path_shape[i]));
  Fr28 s0 = Fr28::zero();  // chain seed s_0 = sum (col - 6) * z[col] over the loaded witnesses
  {
    const Fr28 w7[7] = {new_id, new_trap, new_null, new_acc_hash, old_id, old_trap, old_acc_hash};
    for (int j = 0; j < 7; j++) s0 = s0 + Fr28::mul_inline(rv_small(j + 1), w7[j]);
    for (int i = 0; i < RV_TREE_HEIGHT; i++)
      if (in.path_shape[i]) s0 = s0 + rv_small(8 + i);
  }
  for (int i = 0; i < RV_TREE_HEIGHT; i++) {
    const Fr28 p = rv_load(in.path[i]);
    out.put(p);
    s0 = s0 + Fr28::mul_inline(rv_small(8 + RV_TREE_HEIGHT + i), p);
  }
  out.put(priv_user);
  s0 = s0 + Fr28::mul_inline(rv_small(8 + 2 * RV_TREE_HEIGHT), priv_user);
  for (int i = 0; i < 4; i++) {
    out.put(acc[i]);
    s0 = s0 + Fr28::mul_inline(rv_small(9 + 2 * RV_TREE_HEIGHT + i), acc[i]);
  }

  // verify_note_circuit(new_note, new_note_hash)
  const Fr28 nn[4] = {new_id, new_trap, new_null, new_acc_hash};
  out.set(4, rv_hash(nn, 4, c, out));
  // old note hash, Merkle path
  const Fr28 on[4] = {old_id, old_trap, old_null, old_acc_hash};
  Fr28 cur = rv_hash(on, 4, c, out);
  for (int i = 0; i < RV_TREE_HEIGHT; i++) {
    const Fr28 sibling = rv_load(in.path[i]);
    const Fr28 sel = rv_is_zero(rv_small(in.path_shape[i]), out);
    Fr28 pair[2];
    pair[0] = rv_select(sibling, cur, sel, out);
    pair[1] = rv_select(cur, sibling, sel, out);
    cur = rv_hash(pair, 2, c, out);
  }
  out.set(5, cur);
  // update_account_circuit
  rv_hash(acc, 4, c, out);
  const Fr28 m0 = rv_is_zero(acc[0] - token, out), m1 = rv_is_zero(acc[2] - token, out);
  const Fr28 d0 = Fr28::mul_inline(m0, amount), d1 = Fr28::mul_inline(m1, amount);
  out.put(d0);
  out.put(d1);
  const Fr28 c0 = op_kind == ZKMI_OP_DEPOSIT ? acc[1] + d0 : acc[1] - d0;
  const Fr28 c1 = op_kind == ZKMI_OP_DEPOSIT ? acc[3] + d1 : acc[3] - d1;
  const bool fit0 = rv_range(c0, out), fit1 = rv_range(c1, out);
  if (!(fit0 && fit1) && status == ZKMI_OK) status = ZKMI_ERR_ACCOUNT_UPDATE;
  const Fr28 cv[4] = {acc[0], c0, acc[2], c1};
  rv_hash(cv, 4, c, out);

  // padding chain s_{k+1} = s_k^2 + s_{k-1}
  Fr28 sp = amount + Fr28::mul_inline(rv_small(2), token) + Fr28::mul_inline(rv_small(3), user) +
            Fr28::mul_inline(rv_small(4), old_null);
  Fr28 sc = s0;
  // Representations are lazy (|v| grows by < 1.5 r per addition and products need |v| < 2^12 r):
  // every 64 steps both running values are renormalised by a multiplication with Montgomery 1.
#pragma unroll 1
  for (uint64_t k = 0; k < K; k++) {
    const Fr28 nx = sc.sqr_inline() + sp;
    out.put(nx);
    sp = sc;
    sc = nx;
    if ((k & 63) == 63) {
      sp = Fr28::mul_inline(sp, Fr28::one());
      sc = Fr28::mul_inline(sc, Fr28::one());
    }
  }
  for (uint32_t i = 0; i < n_free; i++) out.put(Fr28::zero());
  return status;
}

}  // namespace zkmi

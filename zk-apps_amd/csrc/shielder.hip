// zkmi — host-side mirror of the reference's prove/verify surface
// (SURVEY.md §8a row a12, §8b): same names, argument meaning and error
// behaviour as shielder/mocked_zk, so callers of mocked_zk::relations::ZkProof
// (contract/lib.rs:56,74; drink_tests/utils/shielder.rs:60,105-114) can be
// pointed at this library one call at a time.
//
//   Scalar               shielder/mocked_zk/src/scalar.rs:1-30
//   Account              shielder/mocked_zk/src/account.rs:10-79
//   Note::hash           shielder/mocked_zk/src/note.rs:25-40
//   Operation::combine   shielder/mocked_zk/src/ops.rs:47-63
//   combine_merkle_hash  shielder/mocked_zk/src/lib.rs:24-28
//   ZkProof              shielder/mocked_zk/src/relations.rs:14-155
// The "knowledge" struct and its SHA-256 checks are reproduced bit-for-bit
// (golden vectors in tests/golden/mock_boundary.json come from the reference's
// own tests).  The Groth16 proof that replaces this mock in production is
// produced by zkmi_groth16_prove; INTEGRATION.md shows how the two connect.
#include <string.h>
#include "../../include/zkmi.h"

namespace {

struct Sha256 {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len = 0;
  Sha256() {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(h, iv, sizeof(iv));
  }
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const uint8_t* p, size_t n) {
    while (n) {
      size_t off = len % 64, take = 64 - off < n ? 64 - off : n;
      memcpy(buf + off, p, take);
      len += take;
      p += take;
      n -= take;
      if (len % 64 == 0) block(buf);
    }
  }
  void finish(uint8_t out[32]) {
    uint64_t bits = len * 8;
    uint8_t pad = 0x80;
    update(&pad, 1);
    uint8_t z = 0;
    while (len % 64 != 56) update(&z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(lb, 8);
    for (int i = 0; i < 8; i++) {
      out[4 * i] = (uint8_t)(h[i] >> 24);
      out[4 * i + 1] = (uint8_t)(h[i] >> 16);
      out[4 * i + 2] = (uint8_t)(h[i] >> 8);
      out[4 * i + 3] = (uint8_t)h[i];
    }
  }
};

bool scalar_eq(const zkmi_scalar& a, const zkmi_scalar& b) { return memcmp(a.bytes, b.bytes, 32) == 0; }

typedef unsigned __int128 u128;
u128 scalar_to_u128(const zkmi_scalar& s) {
  u128 v = 0;
  for (int i = 15; i >= 0; i--) v = (v << 8) | s.bytes[i];
  return v;
}
zkmi_scalar scalar_from_u128(u128 v) {
  zkmi_scalar s;
  memset(s.bytes, 0, 32);
  for (int i = 0; i < 16; i++) s.bytes[i] = (uint8_t)(v >> (8 * i));
  return s;
}

zkmi_scalar account_hash(const zkmi_account& a) {
  // account.rs:16-24 hashes only balances[i].1 for i in 1..TOKENS_NUMBER, each
  // call overwriting the previous digest
  zkmi_scalar res;
  memset(res.bytes, 0, 32);
  for (int i = 1; i < ZKMI_TOKENS_NUMBER; i++) {
    Sha256 h;
    h.update(a.balances[i][1].bytes, 32);
    h.finish(res.bytes);
  }
  return res;
}

zkmi_scalar note_hash(const zkmi_scalar& id, const zkmi_scalar& trapdoor, const zkmi_scalar& nullifier,
                      const zkmi_scalar& acc_hash) {
  Sha256 h;
  h.update(id.bytes, 32);
  h.update(trapdoor.bytes, 32);
  h.update(nullifier.bytes, 32);
  h.update(acc_hash.bytes, 32);
  zkmi_scalar out;
  h.finish(out.bytes);
  return out;
}

int32_t account_update(const zkmi_account& a, const zkmi_op_pub& op, zkmi_account* out) {
  u128 amount = 0;
  for (int i = 15; i >= 0; i--) amount = (amount << 8) | op.amount[i];
  for (int i = 0; i < ZKMI_TOKENS_NUMBER; i++) {
    if (scalar_eq(a.balances[i][0], op.token)) {
      u128 bal = scalar_to_u128(a.balances[i][1]);
      u128 upd;
      if (op.kind == 0) {
        upd = bal + amount;
        if (upd < bal) return ZKMI_ERR_ACCOUNT_UPDATE;  // checked_add
      } else {
        if (amount > bal) return ZKMI_ERR_ACCOUNT_UPDATE;  // checked_sub
        upd = bal - amount;
      }
      *out = a;
      out->balances[i][1] = scalar_from_u128(upd);
      return ZKMI_OK;
    }
  }
  return ZKMI_ERR_ACCOUNT_UPDATE;
}

}  // namespace

extern "C" {

int32_t zkmi_scalar_from_u128(const uint8_t le16[16], zkmi_scalar* out) {
  if (!le16 || !out) return ZKMI_ERR_BAD_ARG;
  memset(out->bytes, 0, 32);
  memcpy(out->bytes, le16, 16);
  return ZKMI_OK;
}
int32_t zkmi_scalar_to_u128(const zkmi_scalar* s, uint8_t out_le16[16]) {
  if (!s || !out_le16) return ZKMI_ERR_BAD_ARG;
  memcpy(out_le16, s->bytes, 16);
  return ZKMI_OK;
}
int32_t zkmi_account_new(const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], zkmi_account* out) {
  if (!tokens || !out) return ZKMI_ERR_BAD_ARG;
  memset(out, 0, sizeof(*out));
  for (int i = 0; i < ZKMI_TOKENS_NUMBER; i++) out->balances[i][0] = tokens[i];
  return ZKMI_OK;
}
int32_t zkmi_account_hash(const zkmi_account* a, zkmi_scalar* out) {
  if (!a || !out) return ZKMI_ERR_BAD_ARG;
  *out = account_hash(*a);
  return ZKMI_OK;
}
int32_t zkmi_operation_combine(const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv) {
  if (!op_pub || !op_priv || op_pub->kind > 1) return ZKMI_ERR_BAD_ARG;
  return scalar_eq(op_pub->user, op_priv->user) ? ZKMI_OK : ZKMI_ERR_OPERATION_COMBINE;
}
int32_t zkmi_account_update(const zkmi_account* a, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv,
                            zkmi_account* out) {
  if (!a || !op_pub || !op_priv || !out || op_pub->kind > 1) return ZKMI_ERR_BAD_ARG;
  return account_update(*a, *op_pub, out);
}
int32_t zkmi_note_hash(const zkmi_scalar* id, const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                       const zkmi_scalar* acc_hash, zkmi_scalar* out) {
  if (!id || !trapdoor || !nullifier || !acc_hash || !out) return ZKMI_ERR_BAD_ARG;
  *out = note_hash(*id, *trapdoor, *nullifier, *acc_hash);
  return ZKMI_OK;
}
int32_t zkmi_combine_merkle_hash(const zkmi_scalar* first, const zkmi_scalar* second, zkmi_scalar* out) {
  if (!first || !second || !out) return ZKMI_ERR_BAD_ARG;
  Sha256 h;
  h.update(first->bytes, 32);
  h.update(second->bytes, 32);
  h.finish(out->bytes);
  return ZKMI_OK;
}

int32_t zkmi_zkproof_new(const zkmi_scalar* id, const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                         const zkmi_op_priv* op_priv, const zkmi_account* acc, zkmi_zkproof* out) {
  if (!id || !trapdoor || !nullifier || !op_priv || !acc || !out) return ZKMI_ERR_BAD_ARG;
  memset(out, 0, sizeof(*out));
  out->id = *id;
  out->trapdoor_new = *trapdoor;
  out->nullifier_new = *nullifier;
  out->acc_new = *acc;
  out->acc_old = *acc;
  out->op_priv = *op_priv;
  return ZKMI_OK;
}

int32_t zkmi_zkproof_update_account(const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv,
                                    const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                                    const zkmi_scalar merkle_proof[ZKMI_MERKLE_TREE_DEPTH], uint32_t leaf_id,
                                    zkmi_scalar* out_h_note_new, zkmi_zkproof* out_new) {
  if (!self || !op_pub || !op_priv || !trapdoor || !nullifier || !merkle_proof || !out_h_note_new || !out_new ||
      op_pub->kind > 1)
    return ZKMI_ERR_BAD_ARG;
  zkmi_account upd;
  int32_t rc = account_update(self->acc_new, *op_pub, &upd);
  if (rc != ZKMI_OK) return rc;
  *out_h_note_new = note_hash(self->id, *trapdoor, *nullifier, account_hash(upd));
  zkmi_zkproof n;
  memset(&n, 0, sizeof(n));
  n.id = self->id;
  n.trapdoor_new = *trapdoor;
  n.trapdoor_old = self->trapdoor_new;
  n.nullifier_new = *nullifier;
  n.acc_new = upd;
  n.acc_old = self->acc_new;
  n.op_priv = *op_priv;
  memcpy(n.merkle_proof, merkle_proof, sizeof(n.merkle_proof));
  n.merkle_proof_leaf_id = leaf_id;
  *out_new = n;
  return ZKMI_OK;
}

int32_t zkmi_zkproof_verify_creation(const zkmi_zkproof* self, const zkmi_scalar* h_note_new,
                                     const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER]) {
  if (!self || !h_note_new || !tokens) return ZKMI_ERR_BAD_ARG;
  zkmi_account acc;
  zkmi_account_new(tokens, &acc);
  zkmi_scalar h = note_hash(self->id, self->trapdoor_new, self->nullifier_new, account_hash(acc));
  return scalar_eq(h, *h_note_new) ? ZKMI_OK : ZKMI_ERR_VERIFICATION;
}

int32_t zkmi_zkproof_verify_update(const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_scalar* h_note_new,
                                   const zkmi_scalar* merkle_root, const zkmi_scalar* nullifier_old) {
  if (!self || !op_pub || !h_note_new || !merkle_root || !nullifier_old || op_pub->kind > 1) return ZKMI_ERR_BAD_ARG;
  const zkmi_scalar h_acc_old = account_hash(self->acc_old);
  if (!scalar_eq(op_pub->user, self->op_priv.user)) return ZKMI_ERR_OPERATION_COMBINE;
  zkmi_account acc_new;
  int32_t rc = account_update(self->acc_old, *op_pub, &acc_new);
  if (rc != ZKMI_OK) return rc;
  const zkmi_scalar h_acc_new = account_hash(acc_new);
  if (!scalar_eq(note_hash(self->id, self->trapdoor_new, self->nullifier_new, h_acc_new), *h_note_new))
    return ZKMI_ERR_VERIFICATION;
  zkmi_scalar cur = note_hash(self->id, self->trapdoor_old, *nullifier_old, h_acc_old);
  uint32_t id = self->merkle_proof_leaf_id;
  for (int i = 0; i < ZKMI_MERKLE_TREE_DEPTH; i++) {
    zkmi_scalar nxt;
    if (id % 2 == 0)
      zkmi_combine_merkle_hash(&cur, &self->merkle_proof[i], &nxt);
    else
      zkmi_combine_merkle_hash(&self->merkle_proof[i], &cur, &nxt);
    cur = nxt;
    id /= 2;
  }
  return scalar_eq(cur, *merkle_root) ? ZKMI_OK : ZKMI_ERR_VERIFICATION;
}

}  // extern "C"

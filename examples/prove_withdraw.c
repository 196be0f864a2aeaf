/* zkmi from plain C: create a note, deposit, withdraw -- proved and verified through include/zkmi.h only.
 *
 *   gcc -O2 -Iinclude examples/prove_withdraw.c -Lzk-apps_amd -lzkmi -Wl,-rpath,$PWD/zk-apps_amd -o prove_withdraw
 *
 * Call for call what a user of mocked_zk::relations::ZkProof does today
 *   ZkProof::new / verify_creation      shielder/contract/drink_tests/utils/shielder.rs:43-76, contract/lib.rs:50-58
 *   update_account / verify_update      shielder/contract/drink_tests/utils/shielder.rs:78-134, contract/lib.rs:63-78
 * with real Groth16 proofs of the Poseidon relations instead of the SHA-256 mock: the caller passes the mock's
 * own types (Scalar, OpPub, OpPriv, ZkProof); the OpPub -> public-input mapping and the choice of key by
 * operation kind happen inside zkmi_shielder_prove_update / zkmi_shielder_verify_update. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkmi.h"

static zkmi_scalar sc(uint64_t v) {
  zkmi_scalar s;
  memset(s.bytes, 0, 32);
  memcpy(s.bytes, &v, 8); /* little-endian host */
  return s;
}

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int32_t rc_ = (call);                                                               \
    if (rc_ != ZKMI_OK) {                                                               \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, ctx ? zkmi_last_error(ctx) : ""); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

/* trusted setup with explicit toxic waste: test / bench use only */
static int32_t setup(zkmi_ctx* ctx, zkmi_r1cs* r1cs, uint8_t seed, zkmi_pk** pk, uint8_t** vk) {
  uint8_t toxic[160];
  for (int i = 0; i < 160; i++) toxic[i] = (uint8_t)(17 * i + seed);
  for (int k = 0; k < 5; k++) toxic[32 * k + 31] &= 0x3f; /* canonical scalars */
  uint32_t n_pub = 0;
  int32_t rc = zkmi_r1cs_shape(r1cs, NULL, &n_pub, NULL, NULL);
  if (rc != ZKMI_OK) return rc;
  const uint64_t cap = 672 + 96 * (uint64_t)n_pub;
  *vk = malloc(cap);
  return zkmi_groth16_setup(ctx, r1cs, toxic, pk, *vk, cap);
}

int main(void) {
  zkmi_ctx* ctx = NULL;
  int32_t rc = zkmi_ctx_create(0, &ctx);
  if (rc != ZKMI_OK) {
    fprintf(stderr, "zkmi_ctx_create -> %d: no gfx950 device; there is no CPU fallback\n", rc);
    return 2;
  }
  /* one key per relation: creation, deposit, withdraw */
  zkmi_r1cs *rc_create = NULL, *rc_dep = NULL, *rc_wd = NULL;
  CHECK(zkmi_create_note_r1cs(12, &rc_create));
  CHECK(zkmi_update_note_r1cs(14, ZKMI_OP_DEPOSIT, &rc_dep));
  CHECK(zkmi_update_note_r1cs(14, ZKMI_OP_WITHDRAW, &rc_wd));
  zkmi_pk *pk_create = NULL, *pk_dep = NULL, *pk_wd = NULL;
  uint8_t *vk_create = NULL, *vk_dep = NULL, *vk_wd = NULL;
  CHECK(setup(ctx, rc_create, 3, &pk_create, &vk_create));
  CHECK(setup(ctx, rc_dep, 5, &pk_dep, &vk_dep));
  CHECK(setup(ctx, rc_wd, 7, &pk_wd, &vk_wd));

  /* the wallet: user 0xA11CE, tokens (7, 9), a fresh note */
  const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER] = {sc(7), sc(9)};
  const zkmi_scalar id = sc(1), trapdoor0 = sc(1000), nullifier0 = sc(2000), user = sc(0xA11CE);
  zkmi_op_priv op_priv;
  op_priv.user = user;
  zkmi_account acc;
  CHECK(zkmi_account_new(tokens, &acc));
  zkmi_zkproof knowledge;
  CHECK(zkmi_zkproof_new(&id, &trapdoor0, &nullifier0, &op_priv, &acc, &knowledge));
  uint8_t r[32] = {5}, s[32] = {9}, proof[192];
  zkmi_scalar h_note;
  CHECK(zkmi_shielder_prove_creation(ctx, pk_create, &knowledge, tokens, r, s, &h_note, proof));
  printf("proof of the note creation: %s\n",
         zkmi_shielder_verify_creation(vk_create, &h_note, tokens, proof) == ZKMI_OK ? "verified" : "REJECTED");

  /* the contract would now add h_note as leaf 0 of its tree; a wallet reads the path back.  Here: a
   * depth-10 path of made-up siblings (the relation recomputes the root it implies and returns it). */
  zkmi_scalar path[ZKMI_MERKLE_TREE_DEPTH];
  for (int i = 0; i < ZKMI_MERKLE_TREE_DEPTH; i++) path[i] = sc(5000 + i);

  /* deposit 1000, then withdraw 250 */
  zkmi_op_pub op;
  memset(&op, 0, sizeof(op));
  op.token = tokens[0];
  op.user = user;
  const uint64_t amounts[2] = {1000, 250};
  zkmi_scalar nullifier_old = nullifier0, root, h_new;
  int ok = 1;
  for (int step = 0; step < 2; step++) {
    op.kind = step == 0 ? ZKMI_OP_DEPOSIT : ZKMI_OP_WITHDRAW;
    memset(op.amount, 0, 16);
    memcpy(op.amount, &amounts[step], 8);
    const zkmi_scalar trapdoor = sc(1001 + step), nullifier = sc(2001 + step);
    zkmi_zkproof next;
    CHECK(zkmi_shielder_prove_update(ctx, pk_dep, pk_wd, &knowledge, &op, &op_priv, &trapdoor, &nullifier, path,
                                     ZKMI_MERKLE_TREE_DEPTH, 0, r, s, &h_new, &root, &next, proof));
    rc = zkmi_shielder_verify_update(vk_dep, vk_wd, &op, &h_new, &root, &nullifier_old, proof);
    printf("proof of the %s: %s\n", step == 0 ? "deposit" : "withdraw", rc == ZKMI_OK ? "verified" : "REJECTED");
    ok = ok && rc == ZKMI_OK;
    op.amount[0] ^= 1; /* another amount */
    printf("same proof, amount tampered: %s\n",
           zkmi_shielder_verify_update(vk_dep, vk_wd, &op, &h_new, &root, &nullifier_old, proof) == ZKMI_ERR_VERIFICATION
               ? "rejected" : "ACCEPTED?!");
    knowledge = next;
    nullifier_old = nullifier;
  }

  /* an impossible update comes back as the mock's ZkpError before any GPU work */
  const uint64_t too_much = 2000;
  memset(op.amount, 0, 16);
  memcpy(op.amount, &too_much, 8);
  const zkmi_scalar t3 = sc(1), n3 = sc(2);
  printf("withdraw above the balance -> %d (ZKMI_ERR_ACCOUNT_UPDATE = %d)\n",
         zkmi_shielder_prove_update(ctx, pk_dep, pk_wd, &knowledge, &op, &op_priv, &t3, &n3, path, ZKMI_MERKLE_TREE_DEPTH, 0, r, s,
                                    &h_new, &root, NULL, proof),
         ZKMI_ERR_ACCOUNT_UPDATE);
  free(vk_create), free(vk_dep), free(vk_wd);
  zkmi_pk_free(pk_create), zkmi_pk_free(pk_dep), zkmi_pk_free(pk_wd);
  zkmi_r1cs_free(rc_create), zkmi_r1cs_free(rc_dep), zkmi_r1cs_free(rc_wd);
  zkmi_ctx_destroy(ctx);
  return ok ? 0 : 1;
}

/* zkmi from plain C: one Shielder withdraw proved and verified through include/zkmi.h only.
 *
 *   gcc -O2 -Iinclude examples/prove_withdraw.c -Lzk-apps_amd -lzkmi -Wl,-rpath,$PWD/zk-apps_amd -o prove_withdraw
 *
 * Mirrors what a caller of mocked_zk::relations::ZkProof::update_account + verify_update does today
 * (shielder/contract/drink_tests/utils/shielder.rs:105-114, shielder/contract/lib.rs:74), with a real
 * Groth16 proof of the update_note relation instead of the SHA-256 mock. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkmi.h"

static void fr_u64(zkmi_fr* f, uint64_t v) {
  memset(f->bytes, 0, 32);
  memcpy(f->bytes, &v, 8); /* little-endian host */
}

#define CHECK(call)                                                            \
  do {                                                                         \
    int32_t rc_ = (call);                                                      \
    if (rc_ != ZKMI_OK) {                                                      \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, ctx ? zkmi_last_error(ctx) : ""); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(void) {
  zkmi_ctx* ctx = NULL;
  const uint32_t log_n = 14;
  int32_t rc = zkmi_ctx_create(0, &ctx);
  if (rc != ZKMI_OK) {
    fprintf(stderr, "zkmi_ctx_create -> %d: no gfx950 device; there is no CPU fallback\n", rc);
    return 2;
  }
  /* relation + keys (trusted setup with explicit toxic waste: test / bench use only) */
  zkmi_r1cs* r1cs = NULL;
  CHECK(zkmi_update_note_r1cs(log_n, ZKMI_OP_WITHDRAW, &r1cs));
  uint8_t toxic[160];
  for (int i = 0; i < 160; i++) toxic[i] = (uint8_t)(17 * i + 3);
  for (int k = 0; k < 5; k++) toxic[32 * k + 31] &= 0x3f; /* canonical scalars */
  uint32_t n_pub = 0;
  CHECK(zkmi_r1cs_shape(r1cs, NULL, &n_pub, NULL, NULL));
  const uint64_t vk_cap = 672 + 96 * (uint64_t)n_pub;
  uint8_t* vk = malloc(vk_cap);
  zkmi_pk* pk = NULL;
  CHECK(zkmi_groth16_setup(ctx, r1cs, toxic, &pk, vk, vk_cap));

  /* the wallet's view of one withdraw: 250 of token 7 out of an account holding (7: 1000, 9: 5) */
  zkmi_note_update in;
  memset(&in, 0, sizeof(in));
  fr_u64(&in.amount, 250);
  fr_u64(&in.token, 7);
  fr_u64(&in.user, 0xA11CE);
  fr_u64(&in.op_priv_user, 0xA11CE);
  fr_u64(&in.new_note[0], 1), fr_u64(&in.new_note[1], 1001), fr_u64(&in.new_note[2], 2001);
  fr_u64(&in.old_note[0], 1), fr_u64(&in.old_note[1], 1000), fr_u64(&in.old_note[2], 2000);
  for (int i = 0; i < 10; i++) {
    in.path_shape[i] = (uint8_t)(i & 1);
    fr_u64(&in.path[i], 5000 + i);
  }
  fr_u64(&in.account[0], 7), fr_u64(&in.account[1], 1000), fr_u64(&in.account[2], 9), fr_u64(&in.account[3], 5);

  uint8_t* z = malloc((size_t)32 << log_n);
  uint8_t publics[6 * 32];
  CHECK(zkmi_update_note_witness(log_n, ZKMI_OP_WITHDRAW, &in, z, publics));
  uint8_t r[32] = {5}, s[32] = {9}, proof[192];
  CHECK(zkmi_groth16_prove(ctx, pk, z, r, s, proof));
  rc = zkmi_groth16_verify(vk, n_pub, publics, proof);
  printf("proof of the withdraw: %s\n", rc == ZKMI_OK ? "verified" : "REJECTED");
  publics[0] ^= 1; /* another amount */
  printf("same proof, amount tampered: %s\n", zkmi_groth16_verify(vk, n_pub, publics, proof) == ZKMI_ERR_VERIFICATION ? "rejected" : "ACCEPTED?!");

  /* an impossible update comes back as the mock's ZkpError */
  fr_u64(&in.amount, 2000);
  printf("withdraw above the balance -> %d (ZKMI_ERR_ACCOUNT_UPDATE = %d)\n",
         zkmi_update_note_witness(log_n, ZKMI_OP_WITHDRAW, &in, z, NULL), ZKMI_ERR_ACCOUNT_UPDATE);
  free(z);
  free(vk);
  zkmi_pk_free(pk);
  zkmi_r1cs_free(r1cs);
  zkmi_ctx_destroy(ctx);
  return rc == ZKMI_OK ? 0 : 1;
}

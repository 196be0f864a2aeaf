// ORACLE / TEST INFRASTRUCTURE — AddressSanitizer + UndefinedBehaviorSanitizer driver for the HOST halves of the
// product (wire encodings, compression, pairing verifier, the mock mirror, the R1CS builders and witness
// generators, the arkworks (de)serialisers, argument validation) and for the C++ oracle.  No GPU is involved
// (GPU AddressSanitizer is not available on this pool): every call below is host-only; zkmi_ctx_create is
// expected to fail with ZKMI_ERR_NO_DEVICE on a CPU box and is exercised for exactly that.
// Build + run: make -C oracle asan
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../include/zkmi.h"
#include "../include/zkmi_testing.h"

extern "C" {
int oracle_ntt_fr(uint8_t* data, uint32_t log_n, int inverse, int coset, int threads);
int oracle_msm_g1(const uint8_t* scalars, const uint8_t* bases, uint64_t n, uint8_t out[96], int threads);
int oracle_msm_g2(const uint8_t* scalars, const uint8_t* bases, uint64_t n, uint8_t out[192], int threads);
}

static int fails = 0;
#define EXPECT(cond)                                              \
  do {                                                            \
    if (!(cond)) {                                                \
      fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
      fails++;                                                    \
    }                                                             \
  } while (0)

static std::vector<uint8_t> hex_field(const std::string& json, const char* key) {
  const std::string k = std::string("\"") + key + "\": \"";
  size_t p = json.find(k);
  std::vector<uint8_t> out;
  if (p == std::string::npos) return out;
  p += k.size();
  auto nib = [](char c) { return (uint8_t)(c <= '9' ? c - '0' : c - 'a' + 10); };
  for (; json[p] != '"'; p += 2) out.push_back((uint8_t)((nib(json[p]) << 4) | nib(json[p + 1])));
  return out;
}

int main(int argc, char** argv) {
  const char* golden = argc > 1 ? argv[1] : "tests/golden/groth16_n128.json";
  std::string json;
  {
    FILE* f = fopen(golden, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", golden); return 2; }
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) json.append(buf, n);
    fclose(f);
  }
  // --- context: no device here -> a clean error, nothing leaked
  zkmi_ctx* ctx = nullptr;
  int32_t rc = zkmi_ctx_create(0, &ctx);
  if (rc == ZKMI_OK) zkmi_ctx_destroy(ctx);
  else EXPECT(rc == ZKMI_ERR_NO_DEVICE && ctx == nullptr);

  // --- wire / compression, valid and malformed
  uint8_t g1[96], g2[192], c1[48], c2[96], t1[96], t2[192];
  EXPECT(zkmi_g1_generator(g1) == 0 && zkmi_g2_generator(g2) == 0);
  EXPECT(zkmi_g1_compress(g1, c1) == 0 && zkmi_g1_decompress(c1, t1) == 0 && !memcmp(g1, t1, 96));
  EXPECT(zkmi_g2_compress(g2, c2) == 0 && zkmi_g2_decompress(c2, t2) == 0 && !memcmp(g2, t2, 192));
  for (int bit = 0; bit < 48 * 8; bit += 7) {  // corrupted encodings: error codes, never a crash
    uint8_t bad[48];
    memcpy(bad, c1, 48);
    bad[bit / 8] ^= (uint8_t)(1u << (bit % 8));
    (void)zkmi_g1_decompress(bad, t1);
  }
  for (int bit = 0; bit < 96 * 8; bit += 13) {
    uint8_t bad[96];
    memcpy(bad, c2, 96);
    bad[bit / 8] ^= (uint8_t)(1u << (bit % 8));
    (void)zkmi_g2_decompress(bad, t2);
  }
  uint8_t k[32] = {0xEE, 0xFF, 0xC0}, fq12[576];
  EXPECT(zkmi_g1_mul(g1, k, t1) == 0 && zkmi_g2_mul(g2, k, t2) == 0);
  EXPECT(zkmi_g1_add(g1, t1, t1) == 0 && zkmi_g2_add(g2, t2, t2) == 0);
  EXPECT(zkmi_g1_in_subgroup(t1) == 0 && zkmi_g2_in_subgroup(t2) == 0);
  EXPECT(zkmi_pairing(t1, t2, fq12) == 0);
  uint8_t big[32];
  memset(big, 0xff, 32);
  EXPECT(zkmi_g1_mul(g1, big, t1) == ZKMI_ERR_NON_CANONICAL);
  uint8_t red[32];
  EXPECT(zkmi_fr_reduce(big, red) == 0);

  // --- verifier on the golden proof, tampered inputs, truncated-looking keys
  std::vector<uint8_t> vk = hex_field(json, "vk"), proof = hex_field(json, "proof"), wit = hex_field(json, "witness");
  EXPECT(vk.size() == 672 + 96 * 7 && proof.size() == 192 && wit.size() >= 32 * 7);
  EXPECT(zkmi_groth16_verify(vk.data(), 7, wit.data() + 32, proof.data()) == ZKMI_OK);
  for (size_t i = 0; i < 192; i += 5) {
    std::vector<uint8_t> p2 = proof;
    p2[i] ^= 0x04;
    EXPECT(zkmi_groth16_verify(vk.data(), 7, wit.data() + 32, p2.data()) != ZKMI_OK);
  }
  // arkworks layout round trip + truncation
  for (int compressed = 0; compressed < 2; compressed++) {
    uint64_t need = 0, used = 0;
    (void)zkmi_ark_vk_write(vk.data(), 7, compressed, nullptr, 0, &need);
    std::vector<uint8_t> blob(need), back(vk.size());
    EXPECT(zkmi_ark_vk_write(vk.data(), 7, compressed, blob.data(), need, &need) == 0);
    uint32_t np = 0;
    EXPECT(zkmi_ark_vk_read(blob.data(), blob.size(), compressed, back.data(), back.size(), &np, &used) == 0 && np == 7 && back == vk);
    for (size_t cut = 0; cut < blob.size(); cut += 37)
      EXPECT(zkmi_ark_vk_read(blob.data(), cut, compressed, back.data(), back.size(), &np, &used) != 0);
    EXPECT(zkmi_ark_vk_read(blob.data(), blob.size(), compressed, back.data(), 100, &np, &used) != 0);  // output too small
  }

  // --- the mock mirror (row a12)
  zkmi_scalar tokens[2], id, trap, null, h;
  memset(tokens, 0, sizeof(tokens));
  memset(tokens[0].bytes, 228, 32);
  uint8_t u7[16] = {7};
  EXPECT(zkmi_scalar_from_u128(u7, &id) == 0);
  trap = id; null = id;
  zkmi_account acc, acc2;
  zkmi_op_priv opp;
  opp.user = id;
  zkmi_zkproof zp, zp2;
  EXPECT(zkmi_account_new(tokens, &acc) == 0 && zkmi_zkproof_new(&id, &trap, &null, &opp, &acc, &zp) == 0);
  EXPECT(zkmi_account_hash(&acc, &h) == 0 && zkmi_note_hash(&id, &trap, &null, &h, &h) == 0);
  EXPECT(zkmi_zkproof_verify_creation(&zp, &h, tokens) == 0);
  zkmi_op_pub op;
  memset(&op, 0, sizeof(op));
  op.kind = 0;
  op.amount[0] = 10;
  op.token = tokens[0];
  op.user = id;
  zkmi_scalar path[ZKMI_MERKLE_TREE_DEPTH];
  memset(path, 0, sizeof(path));
  EXPECT(zkmi_zkproof_update_account(&zp, &op, &opp, &trap, &null, path, 0, &h, &zp2) == 0);
  op.kind = 1;
  op.amount[0] = 200;
  EXPECT(zkmi_account_update(&zp2.acc_new, &op, &opp, &acc2) == ZKMI_ERR_ACCOUNT_UPDATE);

  // --- relations: builders, witness generators, evaluation
  zkmi_r1cs* r = nullptr;
  EXPECT(zkmi_update_note_r1cs_h(13, ZKMI_OP_WITHDRAW, 4, &r) == 0);
  zkmi_note_update in;
  memset(&in, 0, sizeof(in));
  in.tree_height = 4;
  in.amount.bytes[0] = 5;
  in.token.bytes[0] = 7;
  in.account[0].bytes[0] = 7;
  in.account[1].bytes[0] = 50;
  in.account[2].bytes[0] = 9;
  std::vector<uint8_t> z((size_t)32 << 13), zv((size_t)32 << 13);
  uint8_t pub[192];
  EXPECT(zkmi_update_note_witness(13, ZKMI_OP_WITHDRAW, &in, z.data(), pub) == 0);
  EXPECT(zkmi_r1cs_is_satisfied(r, z.data()) == 0);
  EXPECT(zkmi_update_note_witness_values_host(13, ZKMI_OP_WITHDRAW, &in, zv.data()) == 0 && z == zv);
  z[32 * 100] ^= 1;
  EXPECT(zkmi_r1cs_is_satisfied(r, z.data()) == ZKMI_ERR_UNSATISFIED);
  in.amount.bytes[0] = 99;
  EXPECT(zkmi_update_note_witness(13, ZKMI_OP_WITHDRAW, &in, z.data(), pub) == ZKMI_ERR_ACCOUNT_UPDATE);
  in.tree_height = 40;
  EXPECT(zkmi_update_note_witness(13, ZKMI_OP_WITHDRAW, &in, z.data(), pub) == ZKMI_ERR_BAD_ARG);
  uint64_t nnz = 0;
  EXPECT(zkmi_r1cs_export(r, 0, nullptr, nullptr, nullptr, &nnz) == 0 && nnz > 0);
  zkmi_r1cs_free(r);
  EXPECT(zkmi_create_note_r1cs(11, &r) == 0);
  zkmi_note_create nc;
  memset(&nc, 0, sizeof(nc));
  nc.tokens[0].bytes[0] = 3;
  std::vector<uint8_t> zc((size_t)32 << 11);
  EXPECT(zkmi_create_note_witness(11, &nc, zc.data(), pub) == 0 && zkmi_r1cs_is_satisfied(r, zc.data()) == 0);
  zkmi_r1cs_free(r);
  // malformed CSR
  {
    uint32_t rp_bad[3] = {0, 2, 1}, rp[3] = {0, 1, 2}, col[2] = {1, 2};
    uint8_t val[64] = {1};
    val[32] = 1;
    zkmi_r1cs* q = nullptr;
    EXPECT(zkmi_r1cs_create(4, 2, 2, rp_bad, col, val, rp, col, val, rp, col, val, &q) == ZKMI_ERR_BAD_ARG);
    EXPECT(zkmi_r1cs_create(4, 2, 2, rp, col, val, rp, col, val, rp, col, val, &q) == 0);
    zkmi_r1cs_free(q);
  }
  uint32_t mism = 1;
  EXPECT(zkmi_selftest_fq28(3, 50, &mism) == 0 && mism == 0);
  EXPECT(zkmi_selftest_assembly(9, 6, &mism) == 0 && mism == 0);
  EXPECT(zkmi_selftest_host_pool(4, 50, &mism) == 0 && mism == 0);
  {
    // the host combination behind the RCCL all-gather (comm.hip) on all-infinity slots at the 2^26 plan's geometry, 8 ranks
    uint32_t lay[8];
    EXPECT(zkmi_msm_exchange_layout(1ull << 26, 8, lay) == 0 && lay[0] == 13 && lay[4] == 192);
    std::vector<uint8_t> slots((size_t)lay[4] * (lay[2] > lay[3] ? lay[2] : lay[3]) * 8, 0);
    uint8_t res[96];
    EXPECT(zkmi_msm_g1_combine_partials(slots.data(), 8, 1ull << 26, 0, res) == 0);
    EXPECT(zkmi_msm_g1_combine_partials(slots.data(), 8, 1ull << 26, 1, res) == 0);
    EXPECT(zkmi_msm_g1_combine_partials(slots.data(), 0, 1ull << 26, 0, res) == ZKMI_ERR_BAD_ARG);
  }
  EXPECT(zkmi_selftest_poseidon(ZKMI_FIELD_BLS12_381_FR, 5, 10, &mism) == 0 && mism == 0);

  // --- the C++ oracle: small NTT round trip and MSMs over the product's generator multiples
  {
    const uint32_t lg = 6, n = 1u << lg;
    std::vector<uint8_t> a(32 * n), b;
    for (uint32_t i = 0; i < 32 * n; i++) a[i] = (uint8_t)(i * 37 + 11);
    for (uint32_t i = 0; i < n; i++) a[32 * i + 31] &= 0x3f;
    b = a;
    oracle_ntt_fr(b.data(), lg, 0, 1, 2);
    oracle_ntt_fr(b.data(), lg, 1, 1, 2);
    EXPECT(a == b);
    std::vector<uint8_t> pts1(96 * n), pts2(192 * n);
    uint8_t s[32] = {0};
    for (uint32_t i = 0; i < n; i++) {
      s[0] = (uint8_t)(i + 1);
      zkmi_g1_mul(g1, s, pts1.data() + 96 * i);
      zkmi_g2_mul(g2, s, pts2.data() + 192 * i);
    }
    memset(pts1.data() + 96 * 5, 0, 96);  // a base at infinity
    uint8_t o1[96], o2[192];
    EXPECT(oracle_msm_g1(a.data(), pts1.data(), n, o1, 3) == 0 && oracle_msm_g2(a.data(), pts2.data(), n, o2, 3) == 0);
    EXPECT(zkmi_g1_in_subgroup(o1) == 0 && zkmi_g2_in_subgroup(o2) == 0);
  }
  printf("asan_driver: %s (%d failed expectations)\n", fails ? "FAILED" : "ok", fails);
  return fails ? 1 : 0;
}

// zkmi — Groth16 keys in arkworks' CanonicalSerialize byte layout (host code).
//
// A maintainer who has ark-groth16 can hand this library the bytes of `ProvingKey::serialize_*` /
// `VerifyingKey::serialize_*` directly, and can load a key this library generated into arkworks
// (integration/ark_fixture does both directions).  The layout is restated from memory of crates that are
// not in the reference tree (oracle/README.md rows 8 and 10: ark-groth16 0.4.0 data_structures.rs,
// ark-serialize 0.4.2, ark-bls12-381 0.4.0 curves/util.rs):
//   VerifyingKey = alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | Vec<gamma_abc_g1>
//   ProvingKey   = vk | beta_g1 | delta_g1 | Vec a_query | Vec b_g1_query | Vec b_g2_query | Vec h_query | Vec l_query
//   Vec<T>       = u64 little-endian length, then the elements
//   points       = zcash-style big-endian encodings (wire.hip): 48 / 96 B compressed, 96 / 192 B uncompressed
// The reference itself (/root/reference) serialises nothing of the kind: its proof object is the SCALE-encoded
// witness struct (shielder/mocked_zk/src/relations.rs:14-26).
#include <string.h>
#include <vector>
#include "ctx.hpp"

using namespace zkmi;

namespace {

struct Reader {
  const uint8_t* p;
  uint64_t left;
  bool compressed, check;
  bool ok = true;
  bool take(uint64_t n, const uint8_t** out) {
    if (!ok || left < n) return ok = false;
    *out = p;
    p += n;
    left -= n;
    return true;
  }
  bool g1(uint8_t wire[96]) {
    const uint8_t* b;
    G1Affine a;
    if (!take(compressed ? 48 : 96, &b)) return false;
    if (!(compressed ? g1_decompress(b, &a) : g1_read_be(b, &a, check))) return ok = false;
    g1_to_wire(a, wire);
    return true;
  }
  bool g2(uint8_t wire[192]) {
    const uint8_t* b;
    G2Affine a;
    if (!take(compressed ? 96 : 192, &b)) return false;
    if (!(compressed ? g2_decompress(b, &a) : g2_read_be(b, &a, check))) return ok = false;
    g2_to_wire(a, wire);
    return true;
  }
  bool len(uint64_t* n) {
    const uint8_t* b;
    if (!take(8, &b)) return false;
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    *n = v;
    return true;
  }
};

struct Writer {
  uint8_t* p;
  uint64_t cap, used = 0;
  bool compressed;
  bool ok = true;
  uint8_t* room(uint64_t n) {
    used += n;
    if (!p || used > cap) {
      ok = false;
      return nullptr;
    }
    return p + used - n;
  }
  void g1(const uint8_t wire[96]) {
    G1Affine a;
    uint8_t* o = room(compressed ? 48 : 96);
    if (!g1_from_wire(wire, &a, false)) ok = false;
    if (o) compressed ? g1_compress(a, o) : g1_write_be(a, o);
  }
  void g2(const uint8_t wire[192]) {
    G2Affine a;
    uint8_t* o = room(compressed ? 96 : 192);
    if (!g2_from_wire(wire, &a, false)) ok = false;
    if (o) compressed ? g2_compress(a, o) : g2_write_be(a, o);
  }
  void len(uint64_t n) {
    uint8_t* o = room(8);
    if (o)
      for (int i = 0; i < 8; i++) o[i] = (uint8_t)(n >> (8 * i));
  }
};

bool read_vk(Reader& r, std::vector<uint8_t>* vk, uint32_t* n_pub) {
  vk->resize(672);
  if (!r.g1(vk->data()) || !r.g2(vk->data() + 96) || !r.g2(vk->data() + 288) || !r.g2(vk->data() + 480)) return false;
  uint64_t n = 0;
  if (!r.len(&n) || n == 0 || n > (1ull << 26)) return r.ok = false;
  // the length prefix is untrusted: it must be covered by the bytes that are actually there before anything is
  // allocated for it (the library is built without exceptions: a failed allocation would abort the caller)
  if (n > r.left / (r.compressed ? 48 : 96)) return r.ok = false;
  vk->resize(672 + 96 * n);
  for (uint64_t i = 0; i < n; i++)
    if (!r.g1(vk->data() + 672 + 96 * i)) return false;
  // arkworks' Validate::Yes also checks the prime-order subgroup; so does this library's verifier.  `check` asks for
  // it here for the handful of points of a verifying key (the queries of a proving key are checked for curve
  // membership only: an r-torsion check of 5 x 2^20 points on the host would take minutes)
  if (r.check) {
    G1Affine a;
    G2Affine b;
    for (uint64_t i = 0; i <= n; i++)
      if (!g1_from_wire(vk->data() + (i ? 672 + 96 * (i - 1) : 0), &a, true) || !g1_in_subgroup(a)) return r.ok = false;
    for (int k = 0; k < 3; k++)
      if (!g2_from_wire(vk->data() + 96 + 192 * k, &b, true) || !g2_in_subgroup(b)) return r.ok = false;
  }
  *n_pub = (uint32_t)n;
  return true;
}

}  // namespace

extern "C" {

int32_t zkmi_ark_vk_read(const uint8_t* buf, uint64_t len, int32_t compressed, uint8_t* out_vk, uint64_t vk_cap,
                         uint32_t* out_n_pub, uint64_t* out_consumed) {
  if (!buf || !out_vk || !out_n_pub) return ZKMI_ERR_BAD_ARG;
  Reader r{buf, len, compressed != 0, true};
  std::vector<uint8_t> vk;
  uint32_t n_pub = 0;
  if (!read_vk(r, &vk, &n_pub)) return ZKMI_ERR_NON_CANONICAL;
  if (vk.size() > vk_cap) return ZKMI_ERR_BAD_ARG;
  memcpy(out_vk, vk.data(), vk.size());
  *out_n_pub = n_pub;
  if (out_consumed) *out_consumed = len - r.left;
  return ZKMI_OK;
}

int32_t zkmi_ark_vk_write(const uint8_t* vk, uint32_t n_pub, int32_t compressed, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  if (!vk || !out_len || n_pub == 0) return ZKMI_ERR_BAD_ARG;
  Writer w{out, cap, 0, compressed != 0};
  w.g1(vk);
  w.g2(vk + 96);
  w.g2(vk + 288);
  w.g2(vk + 480);
  w.len(n_pub);
  for (uint32_t i = 0; i < n_pub; i++) w.g1(vk + 672 + 96ull * i);
  *out_len = w.used;  // the size needed, also when `out` is NULL or too small
  return w.ok ? ZKMI_OK : ZKMI_ERR_BAD_ARG;
}

int32_t zkmi_ark_pk_load(zkmi_ctx* ctx, const zkmi_r1cs* r1cs, const uint8_t* buf, uint64_t len, int32_t compressed,
                         int32_t check_curve, zkmi_pk** out_pk, uint8_t* out_vk, uint64_t vk_cap) {
  ZK_ENTER(ctx);
  if (!r1cs || !buf || !out_pk) return ZKMI_ERR_BAD_ARG;
  uint32_t n_vars = 0, n_pub = 0, nc = 0, log_n = 0;
  int32_t rc = zkmi_r1cs_shape(r1cs, &n_vars, &n_pub, &nc, &log_n);
  if (rc != ZKMI_OK) return rc;
  Reader r{buf, len, compressed != 0, check_curve != 0};
  std::vector<uint8_t> vk;
  uint32_t vk_pub = 0;
  if (!read_vk(r, &vk, &vk_pub)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "arkworks key: verifying key part");
  if (vk_pub != n_pub) return ctx->fail(ZKMI_ERR_BAD_ARG, "arkworks key: number of instance variables differs from the relation");
  uint8_t beta_g1[96], delta_g1[96];
  if (!r.g1(beta_g1) || !r.g1(delta_g1)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "arkworks key: beta_g1 / delta_g1");
  const uint64_t N = 1ull << log_n;
  const uint64_t want[5] = {n_vars, n_vars, n_vars, N - 1, n_vars - n_pub};
  const bool is_g2[5] = {false, false, true, false, false};
  std::vector<uint8_t> q[5];
  for (int k = 0; k < 5; k++) {
    uint64_t n = 0;
    if (!r.len(&n)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "arkworks key: truncated");
    if (n != want[k]) return ctx->fail(ZKMI_ERR_BAD_ARG, "arkworks key: query length differs from the relation's shape");
    const uint64_t w = is_g2[k] ? 192 : 96;
    if (n > r.left / (compressed ? w / 2 : w)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "arkworks key: truncated");
    q[k].resize(w * n);
    for (uint64_t i = 0; i < n; i++)
      if (!(is_g2[k] ? r.g2(q[k].data() + w * i) : r.g1(q[k].data() + w * i)))
        return ctx->fail(ZKMI_ERR_NON_CANONICAL, "arkworks key: query point");
  }
  if (out_vk) {
    if (vk.size() > vk_cap) return ZKMI_ERR_BAD_ARG;
    memcpy(out_vk, vk.data(), vk.size());
  }
  return zkmi_pk_load(ctx, r1cs, vk.data(), beta_g1, vk.data() + 96, delta_g1, vk.data() + 480, q[0].data(), q[1].data(),
                      q[2].data(), q[3].data(), q[4].data(), out_pk);
}

int32_t zkmi_ark_pk_write(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* vk, int32_t compressed, uint8_t* out,
                          uint64_t cap, uint64_t* out_len) {
  ZK_ENTER(ctx);
  if (!pk || !vk || !out_len) return ZKMI_ERR_BAD_ARG;
  uint32_t n_vars = 0, n_pub = 0, log_n = 0;
  int32_t rc = zkmi_pk_shape(pk, &n_vars, &n_pub, &log_n);
  if (rc != ZKMI_OK) return rc;
  Writer w{out, cap, 0, compressed != 0};
  w.g1(vk);
  w.g2(vk + 96);
  w.g2(vk + 288);
  w.g2(vk + 480);
  w.len(n_pub);
  for (uint32_t i = 0; i < n_pub; i++) w.g1(vk + 672 + 96ull * i);
  uint8_t beta_g1[96], delta_g1[96];
  if ((rc = zkmi_pk_export_g1_elems(pk, beta_g1, delta_g1)) != ZKMI_OK) return rc;
  w.g1(beta_g1);
  w.g1(delta_g1);
  const uint64_t N = 1ull << log_n;
  const uint64_t cnt[5] = {n_vars, n_vars, n_vars, N - 1, n_vars - n_pub};
  const uint64_t CH = 1u << 14;
  std::vector<uint8_t> chunk(192 * CH);
  for (int k = 0; k < 5; k++) {
    const uint64_t wd = k == 2 ? 192 : 96;
    w.len(cnt[k]);
    if (!w.p) {  // size query
      w.used += cnt[k] * (w.compressed ? wd / 2 : wd);
      continue;
    }
    for (uint64_t first = 0; first < cnt[k]; first += CH) {
      const uint64_t c = cnt[k] - first < CH ? cnt[k] - first : CH;
      if ((rc = zkmi_pk_export_query(ctx, pk, k, first, c, chunk.data())) != ZKMI_OK) return rc;
      for (uint64_t i = 0; i < c; i++) k == 2 ? w.g2(chunk.data() + 192 * i) : w.g1(chunk.data() + 96 * i);
    }
  }
  *out_len = w.used;
  return w.ok ? ZKMI_OK : ZKMI_ERR_BAD_ARG;
}

}  // extern "C"

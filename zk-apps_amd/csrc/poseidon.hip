// zkmi — Poseidon-5: constant generation (host), batched hashing and Merkle trees (device),
// C ABI (include/zkmi.h "Poseidon").  SURVEY.md §8f-1.
//
// Constants follow the published procedure of the Poseidon paper's reference generator, which
// is what pse-poseidon 0.2.0 `Spec::new` / halo2-base `OptimizedPoseidonSpec::new::<8, 56, 0>`
// run (crates not in the tree; call site shielder/relations/src/relations/update_note.rs:115-116):
//   Grain LFSR, 80-bit state = field tag 1 (2 bits) | s-box tag 0 (4) | NUM_BITS (12) | t (12) |
//   R_F (10) | R_P (10) | thirty 1s, all MSB first; 160 clocks discarded; output bits taken in
//   pairs (leading 1 emits the second bit); field elements MSB first, round constants by
//   rejection, then 2t elements reduced mod p (no rejection) as x_i, y_j of the Cauchy matrix
//   M[i][j] = 1 / (x_i + y_j).
// oracle/poseidon.py restates the same procedure and pins it against published BN254 vectors.
#include <string.h>
#include <mutex>
#include <vector>
#include "ctx.hpp"
#include "poseidon.hpp"

namespace zkmi {
namespace {

struct Grain {
  uint8_t s[80];
  int head = 0;  // index of b[i]
  Grain(uint32_t nbits, uint32_t t, uint32_t r_f, uint32_t r_p) {
    int pos = 0;
    auto put = [&](uint32_t value, int width) {
      for (int i = 0; i < width; i++) s[pos++] = (uint8_t)((value >> (width - 1 - i)) & 1u);
    };
    put(1, 2);
    put(0, 4);
    put(nbits, 12);
    put(t, 12);
    put(r_f, 10);
    put(r_p, 10);
    put(0x3fffffffu, 30);
    for (int i = 0; i < 160; i++) clock();
  }
  uint8_t clock() {
    auto at = [&](int k) { return s[(head + k) % 80]; };
    const uint8_t b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
    s[head] = b;  // b[i] leaves, b[i+80] enters at the same ring slot
    head = (head + 1) % 80;
    return b;
  }
  uint8_t bit() {
    for (;;) {
      const uint8_t first = clock(), second = clock();
      if (first) return second;
    }
  }
  // nbits-bit integer, most significant bit first, as 8 little-endian words
  void integer(uint32_t nbits, uint32_t w[8]) {
    for (int i = 0; i < 8; i++) w[i] = 0;
    for (int i = (int)nbits - 1; i >= 0; i--)
      if (bit()) w[i >> 5] |= 1u << (i & 31);
  }
};

template <class P>
bool words_lt_mod(const uint32_t w[8]) {
  for (int i = 7; i >= 0; i--)
    if (w[i] != P::MOD32[i]) return w[i] < P::MOD32[i];
  return false;
}
template <class P>
void words_sub_mod(uint32_t w[8]) {
  uint64_t borrow = 0;
  for (int i = 0; i < 8; i++) {
    const uint64_t d = (uint64_t)w[i] - P::MOD32[i] - borrow;
    w[i] = (uint32_t)d;
    borrow = (d >> 63) & 1;
  }
}

template <class P>
struct Spec {
  std::vector<uint8_t> rc, mds;  // canonical bytes
  PoseidonConsts<Fp28<P>> consts;
  Spec() {
    using F = Fp28<P>;
    Grain g(P::NUM_BITS, POS_T, POS_RF, POS_RP);
    rc.resize(32 * POS_ROUNDS * POS_T);
    mds.resize(32 * POS_T * POS_T);
    uint32_t w[8];
    for (int k = 0; k < POS_ROUNDS * POS_T; k++) {
      do g.integer(P::NUM_BITS, w);
      while (!words_lt_mod<P>(w));
      memcpy(rc.data() + 32 * k, w, 32);
      consts.rc[k] = F::from_canonical(w);
    }
    F xy[2 * POS_T];
    for (;;) {
      uint32_t v[2 * POS_T][8];
      for (int k = 0; k < 2 * POS_T; k++) {
        g.integer(P::NUM_BITS, v[k]);
        while (!words_lt_mod<P>(v[k])) words_sub_mod<P>(v[k]);
      }
      bool distinct = true;
      for (int a = 0; a < 2 * POS_T; a++)
        for (int b = a + 1; b < 2 * POS_T; b++)
          if (memcmp(v[a], v[b], 32) == 0) distinct = false;
      if (!distinct) continue;
      for (int k = 0; k < 2 * POS_T; k++) xy[k] = F::from_canonical(v[k]);
      break;
    }
    for (int i = 0; i < POS_T; i++)
      for (int j = 0; j < POS_T; j++) {
        const F m = (xy[i] + xy[POS_T + j]).inv();
        consts.mds[POS_T * i + j] = m;
        m.to_canonical(w);
        memcpy(mds.data() + 32 * (POS_T * i + j), w, 32);
      }
    uint32_t cap[8] = {0, 0, 1, 0, 0, 0, 0, 0};  // 2^64
    consts.cap = F::from_canonical(cap);
  }
};

template <class P>
const Spec<P>& spec() {
  static const Spec<P> s;  // thread-safe one-time construction
  return s;
}

// ---- device ---------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void k_poseidon_hash(const uint32_t* __restrict__ in, uint64_t n, uint32_t arity,
                                                       uint32_t* __restrict__ out,
                                                       const PoseidonConsts<F>* __restrict__ c) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F h = poseidon_hash_words<F>(in + 8ull * arity * i, arity, c);
  uint32_t w[8];
  h.to_canonical(w);
  uint4* o = reinterpret_cast<uint4*>(out + 8ull * i);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F>
hipError_t hash_batch(zkmi_ctx* ctx, int field, const void* d_in, uint64_t n, uint32_t arity, void* d_out,
                      const PoseidonConsts<F>& host_consts) {
  if (!ctx->d_pos[field]) {
    hipError_t e = hipMalloc(&ctx->d_pos[field], sizeof(PoseidonConsts<F>));
    if (e != hipSuccess) return e;
    e = hipMemcpy(ctx->d_pos[field], &host_consts, sizeof(PoseidonConsts<F>), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
  }
  if (n == 0) return hipSuccess;
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(k_poseidon_hash<F>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_in), n,
                     arity, static_cast<uint32_t*>(d_out), static_cast<const PoseidonConsts<F>*>(ctx->d_pos[field]));
  return hipGetLastError();
}

hipError_t hash_batch_any(zkmi_ctx* ctx, int field, const void* d_in, uint64_t n, uint32_t arity, void* d_out) {
  return field == ZKMI_FIELD_BLS12_381_FR
             ? hash_batch<Fr28>(ctx, field, d_in, n, arity, d_out, spec<Fr28Params>().consts)
             : hash_batch<BnFr28>(ctx, field, d_in, n, arity, d_out, spec<BnFr28Params>().consts);
}

}  // namespace

const uint8_t* poseidon_rc_canonical(int field) {
  return field == ZKMI_FIELD_BLS12_381_FR ? spec<Fr28Params>().rc.data() : spec<BnFr28Params>().rc.data();
}
const uint8_t* poseidon_mds_canonical(int field) {
  return field == ZKMI_FIELD_BLS12_381_FR ? spec<Fr28Params>().mds.data() : spec<BnFr28Params>().mds.data();
}
const PoseidonConsts<Fr28>* poseidon_consts_bls() { return &spec<Fr28Params>().consts; }
const PoseidonConsts<BnFr28>* poseidon_consts_bn() { return &spec<BnFr28Params>().consts; }

}  // namespace zkmi

using namespace zkmi;

static bool field_ok(int32_t f) { return f == ZKMI_FIELD_BLS12_381_FR || f == ZKMI_FIELD_BN254_FR; }

extern "C" {

int32_t zkmi_poseidon_spec(int32_t field, uint8_t* out_rc, uint8_t* out_mds) {
  if (!field_ok(field)) return ZKMI_ERR_BAD_ARG;
  if (out_rc) memcpy(out_rc, poseidon_rc_canonical(field), 32 * POS_ROUNDS * POS_T);
  if (out_mds) memcpy(out_mds, poseidon_mds_canonical(field), 32 * POS_T * POS_T);
  return ZKMI_OK;
}

int32_t zkmi_poseidon_hash_batch_dev(zkmi_ctx* ctx, int32_t field, const void* d_in, uint64_t n_hashes, uint32_t arity,
                                     void* d_out) {
  ZK_ENTER(ctx);
  if (!field_ok(field) || arity > 64 || (n_hashes && (!d_out || (arity && !d_in)))) return ZKMI_ERR_BAD_ARG;
  if (ctx->timer()) ctx->timer()->begin(PH_WITNESS, ctx->stream);
  hipError_t e = hash_batch_any(ctx, field, d_in, n_hashes, arity, d_out);
  if (ctx->timer()) ctx->timer()->end(PH_WITNESS, ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "poseidon hash");
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int32_t zkmi_poseidon_hash_batch(zkmi_ctx* ctx, int32_t field, const uint8_t* in, uint64_t n_hashes, uint32_t arity,
                                 uint8_t* out) {
  ZK_ENTER(ctx);
  if (!field_ok(field) || arity > 64 || (n_hashes && (!out || (arity && !in)))) return ZKMI_ERR_BAD_ARG;
  const uint64_t n_in = n_hashes * arity;
  const uint32_t* mod = field == ZKMI_FIELD_BLS12_381_FR ? Fr28Params::MOD32 : BnFr28Params::MOD32;
  for (uint64_t i = 0; i < n_in; i++) {
    uint32_t w[8];
    memcpy(w, in + 32 * i, 32);
    bool lt = false;
    for (int k = 7; k >= 0; k--)
      if (w[k] != mod[k]) {
        lt = w[k] < mod[k];
        break;
      }
    if (!lt) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "poseidon input >= field modulus");
  }
  ZK_HIP(ctx, ctx->staging(32 * (n_in + n_hashes) + 64));
  uint8_t* d_in = static_cast<uint8_t*>(ctx->d_tmp);
  uint8_t* d_out = d_in + 32 * n_in;
  if (n_in) ZK_HIP(ctx, hipMemcpyAsync(d_in, in, 32 * n_in, hipMemcpyHostToDevice, ctx->stream));
  hipError_t e = hash_batch_any(ctx, field, d_in, n_hashes, arity, d_out);
  if (e != hipSuccess) return ctx->hip_fail(e, "poseidon hash");
  if (n_hashes) ZK_HIP(ctx, hipMemcpyAsync(out, d_out, 32 * n_hashes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// All levels of the binary Poseidon tree: d_nodes holds 2 * 2^log_leaves - 1 elements, the leaves
// first (already written by the caller), then each level up to the root at the last position.
int32_t zkmi_poseidon_merkle_tree_dev(zkmi_ctx* ctx, int32_t field, void* d_nodes, uint32_t log_leaves) {
  ZK_ENTER(ctx);
  if (!field_ok(field) || !d_nodes || log_leaves > 30) return ZKMI_ERR_BAD_ARG;
  uint8_t* level = static_cast<uint8_t*>(d_nodes);
  if (ctx->timer()) ctx->timer()->begin(PH_WITNESS, ctx->stream);
  for (uint32_t lv = 0; lv < log_leaves; lv++) {
    const uint64_t n_cur = 1ull << (log_leaves - lv);
    uint8_t* next = level + 32 * n_cur;
    hipError_t e = hash_batch_any(ctx, field, level, n_cur / 2, 2, next);  // pairs are contiguous
    if (e != hipSuccess) return ctx->hip_fail(e, "poseidon tree level");
    level = next;
  }
  if (ctx->timer()) ctx->timer()->end(PH_WITNESS, ctx->stream);
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

}  // extern "C"

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
#include <stdlib.h>
#include <type_traits>
#include <utility>
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
    optimize();
  }

  // Sparse form of the partial rounds.  With S = x^5 on lane 0 only, round i is x -> M S(x + c_i).
  //  (1) constants: lanes 1..4 of c_i pass through S unchanged, so M S(x + c_i) =
  //      M S(x + c_i[0] e0) + M (0, c_i[1..4]); the second term joins the next round's constants.
  //      Going forward leaves one scalar per round and a remainder for the following full round.
  //  (2) matrices: A = [[a, v], [w, B]] factors as [[1, 0], [0, B]] * [[a, v], [B^-1 w, I]]; the
  //      left factor fixes lane 0, commutes with S and with lane-0 constants, and is absorbed into
  //      the next round's matrix M * [[1, 0], [0, B]].  The last left factor is applied once.
  using F = Fp28<P>;
  static void inv4(const F in[16], F out[16]) {
    F a[4][8];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        a[i][j] = in[4 * i + j];
        a[i][4 + j] = i == j ? F::one() : F::zero();
      }
    auto is0 = [](const F& x) {
      uint32_t w[8];
      x.to_canonical(w);
      uint32_t acc = 0;
      for (int k = 0; k < 8; k++) acc |= w[k];
      return acc == 0;
    };
    for (int col = 0; col < 4; col++) {
      int piv = col;
      while (piv < 4 && is0(a[piv][col])) piv++;
      if (piv == 4) abort();  // singular block: cannot happen for a Cauchy-derived chain in practice
      for (int j = 0; j < 8; j++) std::swap(a[col][j], a[piv][j]);
      const F pinv = a[col][col].inv();
      for (int j = 0; j < 8; j++) a[col][j] = a[col][j] * pinv;
      for (int i = 0; i < 4; i++) {
        if (i == col) continue;
        const F f = a[i][col];
        for (int j = 0; j < 8; j++) a[i][j] = a[i][j] - f * a[col][j];
      }
    }
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) out[4 * i + j] = a[i][4 + j];
  }
  void optimize() {
    const F* M = consts.mds;
    const int first = POS_RF / 2;
    F d[POS_T];
    for (int j = 0; j < POS_T; j++) d[j] = consts.rc[POS_T * first + j];
    for (int i = 0; i < POS_RP; i++) {
      consts.pk[i] = d[0];
      F push[POS_T];
      for (int r = 0; r < POS_T; r++) {
        F acc = F::zero();
        for (int j = 1; j < POS_T; j++) acc = acc + M[POS_T * r + j] * d[j];
        push[r] = acc;
      }
      for (int j = 0; j < POS_T; j++) d[j] = consts.rc[POS_T * (first + i + 1) + j] + push[j];
    }
    for (int j = 0; j < POS_T; j++) consts.rc_tail[j] = d[j];
    F A[POS_T * POS_T];
    for (int k = 0; k < POS_T * POS_T; k++) A[k] = M[k];
    for (int i = 0; i < POS_RP; i++) {
      F B[16], Binv[16];
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) B[4 * r + c] = A[POS_T * (r + 1) + (c + 1)];
      inv4(B, Binv);
      for (int j = 0; j < POS_T; j++) consts.prow[POS_T * i + j] = A[j];
      for (int r = 0; r < 4; r++) {
        F acc = F::zero();
        for (int c = 0; c < 4; c++) acc = acc + Binv[4 * r + c] * A[POS_T * (c + 1)];
        consts.pcol[4 * i + r] = acc;
      }
      if (i == POS_RP - 1)
        for (int k = 0; k < 16; k++) consts.plast[k] = B[k];
      // A <- M * [[1, 0], [0, B]]
      F nA[POS_T * POS_T];
      for (int r = 0; r < POS_T; r++) {
        nA[POS_T * r] = M[POS_T * r];
        for (int c = 0; c < 4; c++) {
          F acc = F::zero();
          for (int k = 0; k < 4; k++) acc = acc + M[POS_T * r + (k + 1)] * B[4 * k + c];
          nA[POS_T * r + (c + 1)] = acc;
        }
      }
      for (int k = 0; k < POS_T * POS_T; k++) A[k] = nA[k];
    }
  }
};

template <class P>
const Spec<P>& spec() {
  static const Spec<P> s;  // thread-safe one-time construction
  return s;
}

// ---- device ---------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256, 3) void k_poseidon_hash(const uint32_t* __restrict__ in, uint64_t n, uint32_t arity,
                                                       uint32_t* __restrict__ out,
                                                       const PoseidonConsts<F>* __restrict__ c) {
  __shared__ int32_t tile[POS_LDS_WORDS];  // full-round state, one column per thread
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const F h = poseidon_hash_words_lds<F>(in + 8ull * arity * i, arity, c, tile + threadIdx.x);
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

// Merkle paths out of a tree laid out level after level (zkmi_poseidon_merkle_tree_dev): thread (i, lv)
// copies the sibling of leaf idx[i] at level lv and writes the selector bit of merkle_proof.rs:38-61
// (shape = 0: the sibling is the LEFT input, i.e. the running node is a right child).
__global__ __launch_bounds__(256) void k_merkle_paths(const uint4* __restrict__ nodes, uint32_t log_leaves,
                                                      const uint32_t* __restrict__ idx, uint32_t n,
                                                      uint8_t* __restrict__ shape, uint4* __restrict__ paths) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * log_leaves) return;
  const uint32_t i = t / log_leaves, lv = t % log_leaves;
  const uint64_t n_leaves = 1ull << log_leaves;
  const uint64_t level_off = 2 * n_leaves - (n_leaves >> (lv ? lv - 1 : 0)) * (lv ? 1 : 2);  // sum_{k<lv} n_leaves >> k
  const uint32_t pos = idx[i] >> lv;
  const uint64_t node = level_off + (pos ^ 1u);
  shape[t] = (uint8_t)(1u - (pos & 1u));
  paths[2 * (uint64_t)t] = nodes[2 * node];
  paths[2 * (uint64_t)t + 1] = nodes[2 * node + 1];
}

// roots[i] = CircuitMerkleProof::verify's running node after `depth` levels (merkle_proof.rs:38-61)
template <class F>
__global__ __launch_bounds__(256, 3) void k_merkle_roots(const uint32_t* __restrict__ leaves,
                                                         const uint8_t* __restrict__ shape,
                                                         const uint32_t* __restrict__ paths, uint32_t depth, uint64_t n,
                                                         uint32_t* __restrict__ roots,
                                                         const PoseidonConsts<F>* __restrict__ c) {
  __shared__ int32_t tile[POS_LDS_WORDS];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t pair[16];
  uint32_t cur[8];
#pragma unroll
  for (int k = 0; k < 8; k++) cur[k] = leaves[8 * i + k];
#pragma unroll 1
  for (uint32_t lv = 0; lv < depth; lv++) {
    const uint32_t* sib = paths + 8 * (i * depth + lv);
    const bool sibling_left = shape[i * depth + lv] == 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t s = sib[k];
      pair[k] = sibling_left ? s : cur[k];
      pair[8 + k] = sibling_left ? cur[k] : s;
    }
    const F h = poseidon_hash_words_lds<F>(pair, 2, c, tile + threadIdx.x);
    h.to_canonical(cur);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) roots[8 * i + k] = cur[k];
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

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
// host self-test: sparse form == plain form of the permutation on random states
int32_t zkmi_selftest_poseidon(int32_t field, uint64_t seed, uint32_t iters, uint32_t* out_mismatches) {
  if (!field_ok(field) || !out_mismatches) return ZKMI_ERR_BAD_ARG;
  uint32_t bad = 0;
  uint64_t x = seed;
  auto next = [&]() {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t v = x;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
  };
  auto run = [&](auto* consts) {
    using F = typename std::remove_const<typename std::remove_reference<decltype(consts->cap)>::type>::type;
    for (uint32_t it = 0; it < iters; it++) {
      F a[POS_T], b[POS_T];
      for (int j = 0; j < POS_T; j++) {
        uint32_t w[8];
        for (int k = 0; k < 8; k += 2) {
          const uint64_t v = next();
          w[k] = (uint32_t)v;
          w[k + 1] = (uint32_t)(v >> 32);
        }
        w[7] &= 0x0fffffffu;  // < 2^252 < modulus
        if (it == 0) memset(w, 0, sizeof(w));
        a[j] = b[j] = F::from_canonical(w);
      }
      poseidon_permute_plain(a, consts);
      poseidon_permute(b, consts);
      for (int j = 0; j < POS_T; j++) {
        uint32_t wa[8], wb[8];
        a[j].to_canonical(wa);
        b[j].to_canonical(wb);
        if (memcmp(wa, wb, 32) != 0) bad++;
      }
    }
  };
  if (field == ZKMI_FIELD_BLS12_381_FR) run(poseidon_consts_bls());
  else run(poseidon_consts_bn());
  *out_mismatches = bad;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

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

// Paths of n leaves out of the node array of zkmi_poseidon_merkle_tree_dev: out_shape = n x log_leaves
// selector bytes, out_paths = n x log_leaves x 32 B siblings (host buffers) -- the MerkleProof inputs of
// zkmi_note_update.
int32_t zkmi_poseidon_merkle_paths_dev(zkmi_ctx* ctx, const void* d_nodes, uint32_t log_leaves, const uint32_t* leaf_idx,
                                       uint32_t n, uint8_t* out_shape, uint8_t* out_paths) {
  ZK_ENTER(ctx);
  if (!d_nodes || log_leaves == 0 || log_leaves > 30 || (n && (!leaf_idx || !out_shape || !out_paths)))
    return ZKMI_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++)
    if (leaf_idx[i] >> log_leaves) return ZKMI_ERR_BAD_ARG;
  if (n == 0) return ZKMI_OK;
  const uint64_t cnt = (uint64_t)n * log_leaves;
  const uint64_t off_shape = ((uint64_t)n * 4 + 255) & ~255ull, off_paths = (off_shape + cnt + 255) & ~255ull;
  ZK_HIP(ctx, ctx->staging(off_paths + 32 * cnt));
  uint8_t* d = static_cast<uint8_t*>(ctx->d_tmp);
  ZK_HIP(ctx, hipMemcpyAsync(d, leaf_idx, (uint64_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_merkle_paths, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const uint4*>(d_nodes), log_leaves, reinterpret_cast<const uint32_t*>(d), n,
                     d + off_shape, reinterpret_cast<uint4*>(d + off_paths));
  ZK_HIP(ctx, hipGetLastError());
  ZK_HIP(ctx, hipMemcpyAsync(out_shape, d + off_shape, cnt, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipMemcpyAsync(out_paths, d + off_paths, 32 * cnt, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// Batch of Merkle-path recomputations on the device (what CircuitMerkleProof::verify constrains):
// d_leaves n x 32 B, d_shape n x depth bytes, d_paths n x depth x 32 B -> d_roots n x 32 B.
int32_t zkmi_poseidon_merkle_roots_dev(zkmi_ctx* ctx, int32_t field, const void* d_leaves, const void* d_shape,
                                       const void* d_paths, uint32_t depth, uint64_t n, void* d_roots) {
  ZK_ENTER(ctx);
  if (!field_ok(field) || depth > 64 || (n && (!d_leaves || !d_roots || (depth && (!d_shape || !d_paths)))))
    return ZKMI_ERR_BAD_ARG;
  if (n == 0) return ZKMI_OK;
  // make sure the constants are resident (hash_batch uploads them on first use)
  hipError_t e = hash_batch_any(ctx, field, nullptr, 0, 2, nullptr);
  if (e != hipSuccess) return ctx->hip_fail(e, "poseidon constants");
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  if (ctx->timer()) ctx->timer()->begin(PH_WITNESS, ctx->stream);
  if (field == ZKMI_FIELD_BLS12_381_FR)
    hipLaunchKernelGGL(k_merkle_roots<Fr28>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_leaves),
                       static_cast<const uint8_t*>(d_shape), static_cast<const uint32_t*>(d_paths), depth, n,
                       static_cast<uint32_t*>(d_roots), static_cast<const PoseidonConsts<Fr28>*>(ctx->d_pos[field]));
  else
    hipLaunchKernelGGL(k_merkle_roots<BnFr28>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(d_leaves),
                       static_cast<const uint8_t*>(d_shape), static_cast<const uint32_t*>(d_paths), depth, n,
                       static_cast<uint32_t*>(d_roots), static_cast<const PoseidonConsts<BnFr28>*>(ctx->d_pos[field]));
  if (ctx->timer()) ctx->timer()->end(PH_WITNESS, ctx->stream);
  ZK_HIP(ctx, hipGetLastError());
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

}  // extern "C"

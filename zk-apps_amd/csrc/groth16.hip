// zkmi — Groth16 over BLS12-381: witness -> proof on the device, setup with
// explicit toxic waste, pairing verifier on the host.
//
// Reference locus: none in /root/reference (SURVEY.md §0, §8a rows a7, a10,
// a11); the call this replaces is the mock prove step
// ZkProof::update_account (shielder/mocked_zk/src/relations.rs:79-98).  The
// algorithm restates ark-groth16 0.4 [not in tree] exactly as oracle/groth16.py
// does: LibsnarkReduction witness map (3 iNTT, 3 coset NTT, pointwise
// quotient, 1 coset iNTT), A/B/C assembly with explicit (r, s).
//
// Device pipeline for one proof (all on the ctx stream, key resident in HBM):
//   upload z (32 B/var) -> to Montgomery
//   k_matvec x3            a,b,c = <A_i,z>, <B_i,z>, <C_i,z>  (+ input rows)
//   NTT x7                 see ntt.hip
//   k_quotient             h = (a*b - c) / Z(g)
//   digit-sort(z[1..])     shared by the A, B1, B2 and L MSMs (L query is
//                          stored padded with n_pub-1 infinities so it lines up)
//   MSM G1 x3, MSM G2 x1, digit-sort(h), MSM G1 (H)
//   host: O(1) scalar multiplications, compression to 192 bytes
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <memory>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>
#include "ctx.hpp"
#include "tune.hpp"
#include "host_pool.hpp"
#include "pairing.hpp"
#include "r1cs.hpp"

namespace zkmi {

template <class T>
__device__ __forceinline__ T ldv(const T* p) {
  T r;
  const uint4* s = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
  return r;
}
template <class T>
__device__ __forceinline__ void stv(T* p, const T& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
}

// the three constraint matrices of a relation (CSR, limb-form values), handed to k_matvec by value
struct MatSet {
  const uint32_t* rowptr[3];
  const uint32_t* col[3];
  const Fr28* val[3];
};
// out[m][i] = <M_i, z> for i < nc; rows [nc, nc+n_pub) = z_j for m = A (0), else 0; rest 0.
// blockIdx.z = matrix (A, B, C in ONE launch: the three mat-vecs, like the three transforms after them, are queued as
// one grid each -- every extra launch on the front stream is one more hand-over between it and the accumulations,
// DESIGN.md 4.10); the outputs of the three matrices lie back to back (stride = proofs x n)
// S lanes per row (adjacent lanes, partial sums added over __shfl_xor): 1 for throughput; 8 for one small proof, whose
// mat-vec lasts as long as its longest row on one lane (the Poseidon relation has rows of ~100 terms: 0.12 ms at the head of
// the latency chain, section 4.11)
// A row with more terms than this is not one lane's work (a linear combination over a whole vector -- a sum of 2^20 bits --
// would keep one lane busy for a second while the chip waits: the rows of a relation are as long as its author made them).
// The key lists such rows at setup (normally none, and then nothing is launched); k_matvec leaves them to k_matvec_long:
// one 256-thread workgroup per (listed row, proof of the group), terms dealt round-robin, partial sums added through LDS.
constexpr uint32_t MATVEC_LONG_ROW = 1024;
__global__ void __launch_bounds__(256)
k_matvec_long(MatSet ms, const uint32_t* __restrict__ rows, const Fr28* __restrict__ z, Fr28* __restrict__ out_base, uint32_t n, uint32_t n_vars) {
  __shared__ Fr28 part[256];
  const uint32_t m = rows[2 * blockIdx.x], i = rows[2 * blockIdx.x + 1];
  const uint32_t* __restrict__ col = ms.col[m];
  const Fr28* __restrict__ val = ms.val[m];
  z += (size_t)blockIdx.y * n_vars;
  Fr28 acc = Fr28::zero();
  const uint32_t b = ms.rowptr[m][i], e = ms.rowptr[m][i + 1];
  for (uint32_t k = b + threadIdx.x; k < e; k += 256) acc = acc + ld28(val + k) * ld28(z + col[k]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] = part[threadIdx.x] + part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) st28(out_base + (size_t)m * gridDim.y * n + (size_t)blockIdx.y * n + i, part[0]);
}

template <int S>
__global__ void __launch_bounds__(256)
k_matvec(MatSet ms, const Fr28* __restrict__ z, Fr28* __restrict__ out_base, uint32_t nc, uint32_t n, uint32_t n_pub,
         uint32_t n_vars) {
  const int m = blockIdx.z;
  const uint32_t* __restrict__ rowptr = ms.rowptr[m];
  const uint32_t* __restrict__ col = ms.col[m];
  const Fr28* __restrict__ val = ms.val[m];
  const int is_a = m == 0;
  Fr28* __restrict__ out = out_base + (size_t)m * gridDim.y * n;
  // blockIdx.y = proof of a group: assignments of n_vars elements and outputs of n elements back to back
  z += (size_t)blockIdx.y * n_vars;
  out += (size_t)blockIdx.y * n;
  const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = gi / S, lane = gi % S;
  if (i >= n) return;  // the S lanes of a row leave together
  Fr28 acc = Fr28::zero();
  if (i < nc) {
    const uint32_t b = rowptr[i], e = rowptr[i + 1];
    if (e - b > MATVEC_LONG_ROW) return;  // k_matvec_long's (all S lanes of the row see the same bounds)
    for (uint32_t k = b + lane; k < e; k += S) acc = acc + ld28(val + k) * ld28(z + col[k]);
  } else if (is_a && i < nc + n_pub && lane == 0) {
    acc = ld28(z + (i - nc));
  }
  if constexpr (S > 1) {
#pragma unroll
    for (int o = S / 2; o > 0; o >>= 1) {
      Fr28 t;
#pragma unroll
      for (int q = 0; q < Fr28::NL; q++) t.l[q] = __shfl_xor(acc.l[q], o);
      acc = acc + t;
    }
    if (lane != 0) return;
  }
  st28(out + i, acc);
}

// Satisfaction check on the evaluations the mat-vecs just produced: a_i b_i = c_i on every constraint
// row and z_0 = 1 (the prover adds the constant column's query entries unconditionally).  A proof
// from an assignment that fails this cannot verify; the prover reports ZKMI_ERR_UNSATISFIED instead.
__global__ void __launch_bounds__(256)
k_check_sat(const Fr28* __restrict__ a, const Fr28* __restrict__ b, const Fr28* __restrict__ c,
            const Fr28* __restrict__ z, uint32_t nc, uint32_t* __restrict__ flag, uint32_t n, uint32_t n_vars) {
  a += (size_t)blockIdx.y * n;
  b += (size_t)blockIdx.y * n;
  c += (size_t)blockIdx.y * n;
  z += (size_t)blockIdx.y * n_vars;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < nc) {
    // one more product by R = one() brings the difference back into is_zero()'s exact range
    const Fr28 d = (ld28(a + i) * ld28(b + i) - ld28(c + i)) * Fr28::one();
    bad = !d.is_zero();
  }
  if (i == 0) {
    const Fr28 d = (ld28(z) - Fr28::one()) * Fr28::one();
    bad = bad || !d.is_zero();
  }
  // (the flag word lives in pinned HOST memory: a rare system-scope atomic instead of a memset kernel in front of and a
  // copy kernel behind every proof's checks -- both were 512-thread blits that waited ~0.2 ms each for SIMDs on the front stream)
  if (bad) __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every witness element must be a canonical integer < r: the MSM digits are taken from the raw words while
// the mat-vec works on residues, so a non-canonical element would make the two halves of the proof disagree
__global__ void __launch_bounds__(256)
k_check_canonical(const uint32_t* __restrict__ z, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(z + (size_t)i * 8);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  bool lt = false, decided = false;
#pragma unroll
  for (int k = 7; k >= 0; k--)
    if (!decided && w[k] != Fr28Params::MOD32[k]) {
      lt = w[k] < Fr28Params::MOD32[k];
      decided = true;
    }
  if (!lt) __hip_atomic_fetch_or(flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256)
k_quotient(Fr28* __restrict__ a, const Fr28* __restrict__ b, const Fr28* __restrict__ c, Fr28 zinv, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st28(a + i, (ld28(a + i) * ld28(b + i) - ld28(c + i)) * zinv);
}

// the device-resident assignments of a group, gathered back to back: ONE launch of one-wave workgroups instead of one
// blit kernel per proof (1 300 __amd_rocclr_copyBuffer launches of 512 threads took 9.6 % of a 2^14 run's kernel time:
// each waited for SIMDs that accumulation waves kept full, section 4.10)
struct ZPtrSet {
  const uint4* p[64];
};
__global__ void __launch_bounds__(64)
k_gather_z(ZPtrSet src, uint4* __restrict__ dst, uint32_t vec16) {  // vec16 = 16-byte words per assignment
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= vec16) return;
  dst[(size_t)blockIdx.y * vec16 + i] = src.p[blockIdx.y][i];
}

// out_i = r * z_i mod the scalar modulus, canonical words in and out: the scalars of the B1 MSM when it is folded into the
// reduction C's other summands share (r * MSM(b1, z) = MSM(b1, r z)).  blockIdx.y = proof of the group: its own r, its
// vector `stride` words further on.
struct RSet {
  Fr28 r[64];
};
__global__ void __launch_bounds__(64)
k_scale_canonical(const uint32_t* __restrict__ in, RSet rs, uint32_t* __restrict__ out, uint32_t n, uint64_t stride) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const size_t at = (size_t)blockIdx.y * stride + (size_t)i * 8;
  const uint4* q = reinterpret_cast<const uint4*>(in + at);
  const uint4 a = q[0], b = q[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const Fr28 v = Fr28::from_canonical(w) * rs.r[blockIdx.y];
  uint32_t o[8];
  v.to_canonical(o);
  uint4* d = reinterpret_cast<uint4*>(out + at);
  d[0] = make_uint4(o[0], o[1], o[2], o[3]);
  d[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// The digit sort of z knows how many NON-ZERO digits the assignment has (entries per partition); their sum goes to a pinned
// host word, from which the prover decides whether the next proofs of this key fold B1 (prove_enqueue_z): r z has all its
// digits whatever z looked like.
__global__ void __launch_bounds__(64)
k_entries_to_host(const uint32_t* __restrict__ part_total, uint32_t np, uint32_t* __restrict__ host_word) {
  uint32_t s = 0;
  for (uint32_t i = threadIdx.x; i < np; i += 64) s += part_total[i];
  for (int o = 32; o; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) __hip_atomic_store(host_word, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// out[p] = in[rev(p)]: bases of the H MSM follow the bit-reversed coefficient order
template <class A>
__global__ void __launch_bounds__(256)
k_bitrev_points(const A* __restrict__ in, A* __restrict__ out, int log_n) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (1u << log_n)) return;
  const uint32_t r = log_n ? (__brev(p) >> (32 - log_n)) : 0u;
  stv(out + p, ldv(in + r));
}

// fixed-base multiplication out[i] = s_i * B with 8-bit windows:
// table[w*256 + d] = d * 2^(8w) * B (affine, Montgomery), w < 32
template <class F>
__global__ void __launch_bounds__(64)
k_fixed_base(const uint32_t* __restrict__ scalars, const Affine<F>* __restrict__ table, Affine<F>* __restrict__ out,
             uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int w = 0; w < 32; w++) {
    const uint32_t limb = scalars[(size_t)i * 8 + (w >> 2)];
    const uint32_t d = (limb >> ((w & 3) * 8)) & 0xffu;
    if (d) {
      Affine<F> p = ldv(table + w * 256 + d);
      acc.madd(p);
    }
  }
  stv(out + i, acc.to_affine());
}

}  // namespace zkmi

using namespace zkmi;

// Scheduling switches of the prover -- A/B library only (tune.hpp); the product library compiles the defaults in.
// Measured in DESIGN.md 4.4 (both are within noise of each other at N = 2^20):
//   ZKMI_SORT_SIDE=1  digit sorts on their own high-priority stream into ping-pong z buffers (default: main stream)
//   ZKMI_AUX_SPLIT=0  all reductions of a proof on one stream (default: three streams)
static bool prover_sort_side() { return ZK_TUNE("ZKMI_SORT_SIDE", 0) == 1; }
// largest domain (log2) whose single proofs take the latency path: h sorted ahead of the accumulations, A, B1, L and H
// accumulated in one launch (ZKMI_SOLO_MAX_LOG).  Measured, same box: 2^17 5.3 vs 5.4 ms, 2^18 9.3 -> 8.9, 2^19 16.2 -> 14.1,
// 2^20 20.9 -> 22.9 (there the z accumulations should not wait for the transforms: the default path starts them at once)
static uint32_t prover_solo_max_log() {
  const int v = ZK_TUNE("ZKMI_SOLO_MAX_LOG", 19);
  return (uint32_t)(v < 0 ? 0 : v > 28 ? 28 : v);
}
static bool prover_aux_split() { return ZK_TUNE("ZKMI_AUX_SPLIT", 1) != 0; }

// two digit sorts whose bucket arrays can be added bucket by bucket: same digit width, partitions and segment length
static bool same_bucket_set(const MsmPlan& a, const MsmPlan& b) {
  return a.shared == b.shared && a.c == b.c && a.nwin == b.nwin && a.nb == b.nb && a.seg_log == b.seg_log &&
         a.vec_parts == b.vec_parts && a.ndigits == b.ndigits;
}

struct zkmi_pk {
  zkmi_ctx* ctx = nullptr;
  int device = -1;
  uint32_t n_vars = 0, n_pub = 0, nc = 0, log_n = 0;
  uint32_t tree_height = 0;  // from the zkmi_r1cs the key was made for (0 = not an update_note relation)
  // small domains: up to `gmax` proofs travel through the pipeline as ONE group (one sort, one accumulation launch
  // per query, batched NTT passes); every per-proof buffer below holds gmax vectors back to back
  uint32_t gmax = 1;
  uint32_t* d_rowptr[3] = {nullptr, nullptr, nullptr};
  uint32_t* d_long_rows = nullptr;  // (matrix, row) of every row with more than MATVEC_LONG_ROW terms (k_matvec_long)
  uint32_t n_long_rows = 0;
  uint32_t* d_col[3] = {nullptr, nullptr, nullptr};
  Fr28* d_val[3] = {nullptr, nullptr, nullptr};
  G1Affine *a_query = nullptr, *b_g1_query = nullptr, *h_query = nullptr, *l_query = nullptr;  // l padded to n_vars
  G2Affine* b_g2_query = nullptr;
  // the same queries in the device MSM representation (28-bit limbs)
  Affine<Fq28>*a28 = nullptr, *b1_28 = nullptr, *h28 = nullptr, *h28_rev = nullptr, *l28 = nullptr;
  Affine<Fq2_28>* b2_28 = nullptr;
  // shared-bucket MSM tables: entry [w * n + i] = 2^(c w) * query[i] (msm_impl.hpp)
  bool shared = false;
  Affine<Fq28>*a_tab = nullptr, *b1_tab = nullptr, *l_tab = nullptr, *h_tab = nullptr;
  Affine<Fq2_28>* b2_tab = nullptr;
  G1Affine alpha_g1, beta_g1, delta_g1, a0, b1_0;
  G2Affine beta_g2, delta_g2, b2_0;
  // host tables of proof assembly (built once per key): r * delta_1, s * delta_2 and r * (beta_1 + b1_0) are multiplications
  // by points the key fixes; the constant summands of A and B in affine form
  std::unique_ptr<FixedBase8<Fq>> delta1_tab, k1_tab;
  std::unique_ptr<FixedBase8<Fq2>> delta2_tab;
  G1Affine ka;   // alpha_1 + a0
  G2Affine kb2;  // beta_2 + b2_0
  // per proof in flight (ring of zkmi_ctx::PROOF_RING): witness in canonical words (digit source of the
  // A/B/L MSMs) and h coefficients in canonical words, bit-reversed order (digit source of the H MSM)
  Fr* d_z[zkmi_ctx::PROOF_RING] = {};
  // where the assignment(s) of the proof (group) in flight in ring slot `par` actually lie: d_z[par], or -- one
  // device-resident assignment -- the caller's own buffer (no copy at all; it stays valid for the duration of the call)
  mutable const Fr* z_cur[zkmi_ctx::PROOF_RING] = {};
  uint32_t* d_rz[zkmi_ctx::PROOF_RING] = {};  // r * z in canonical words, laid out like d_z: the scalars of the B1 MSM (prove_enqueue_z)
  uint32_t* d_h[zkmi_ctx::PROOF_RING] = {};
  Fr28 *d_zm = nullptr, *d_a = nullptr;  // limb form (field28.hpp), front stream only; d_a holds a, b, c of a group back to back
  // per proof in flight, in pinned host memory: bit 0 set by k_check_sat, bit 1 by k_check_canonical (system-scope atomics);
  // valid once the proof's H MSM has landed (it waits for the front stream), cleared by the host when it has read it
  uint32_t* h_unsat = nullptr;
  // non-zero digits of the assignment(s) of the proof (group) in ring slot i, written by k_entries_to_host behind the digit
  // sort of z (same pinned allocation, words RING..2 RING); and what prove_finish made of the last one it read: B1 is only
  // folded into the L + H reduction (its MSM taken over r z, all of whose digits are non-zero) while the assignments of this
  // key fill at least 9 in 10 of their digits -- a witness of bits is cheaper through the sort of z itself (106.8 against 68.4
  // proofs/s for 2^20 bit constraints: profiles/r04/experiments/rb1_fold_ab.txt).  The proof bytes do not depend on it.
  uint32_t* h_zent = nullptr;
  mutable bool fold_dense = true;
  mutable uint64_t last_entries = 0, last_full = 0;  // what note_density() last read / compared it with (zkmi_pk_schedule_state)
  ~zkmi_pk() {
    if (device >= 0) (void)hipSetDevice(device);  // the key's buffers live on its context's device (the ctx may be gone)
    (void)hipDeviceSynchronize();  // nothing queued by an earlier call may still read the key or write its pinned flags
    for (int m = 0; m < 3; m++) {
      if (d_rowptr[m]) (void)hipFree(d_rowptr[m]);
      if (d_col[m]) (void)hipFree(d_col[m]);
      if (d_val[m]) (void)hipFree(d_val[m]);
    }
    if (d_long_rows) (void)hipFree(d_long_rows);
    void* ptrs[] = {a_query, b_g1_query, h_query, l_query, b_g2_query, d_zm, d_a, a28, b1_28, h28, h28_rev, l28, b2_28,
                    a_tab, b1_tab, l_tab, h_tab, b2_tab, d_z[0], d_z[1], d_z[2], d_h[0], d_h[1], d_h[2]};
    static_assert(zkmi_ctx::PROOF_RING == 3, "ring size");
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    if (h_unsat) (void)hipHostFree(h_unsat);
    for (uint32_t* p : d_rz)
      if (p) (void)hipFree(p);
  }
};

static hipError_t pk_alloc(zkmi_pk* pk, zkmi_ctx* ctx, const zkmi_r1cs* r) {
  pk->ctx = ctx;
  pk->device = ctx->device;
  pk->n_vars = r->n_vars;
  pk->n_pub = r->n_pub;
  pk->nc = r->n_constraints;
  pk->log_n = r->log_n;
  pk->tree_height = r->tree_height;
  const uint64_t N = 1ull << r->log_n;
  hipError_t e;
  {
    std::vector<uint32_t> longs;
    for (uint32_t m = 0; m < 3; m++)
      for (uint32_t i = 0; i + 1 < r->m[m].rowptr.size(); i++)
        if (r->m[m].rowptr[i + 1] - r->m[m].rowptr[i] > MATVEC_LONG_ROW) {
          longs.push_back(m);
          longs.push_back(i);
        }
    pk->n_long_rows = (uint32_t)(longs.size() / 2);
    if (pk->n_long_rows) {
      if ((e = hipMalloc(&pk->d_long_rows, sizeof(uint32_t) * longs.size())) != hipSuccess) return e;
      if ((e = hipMemcpy(pk->d_long_rows, longs.data(), sizeof(uint32_t) * longs.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
  }
  for (int m = 0; m < 3; m++) {
    const auto& c = r->m[m];
    const size_t nnz = c.col.size();
    if ((e = hipMalloc(&pk->d_rowptr[m], sizeof(uint32_t) * c.rowptr.size())) != hipSuccess) return e;
    if ((e = hipMalloc(&pk->d_col[m], sizeof(uint32_t) * (nnz ? nnz : 1))) != hipSuccess) return e;
    if ((e = hipMalloc(&pk->d_val[m], sizeof(Fr28) * (nnz ? nnz : 1))) != hipSuccess) return e;
    if ((e = hipMemcpy(pk->d_rowptr[m], c.rowptr.data(), sizeof(uint32_t) * c.rowptr.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if (nnz) {
      if ((e = hipMemcpy(pk->d_col[m], c.col.data(), sizeof(uint32_t) * nnz, hipMemcpyHostToDevice)) != hipSuccess) return e;
      // matrix values: canonical words on the host -> limb form on the device
      std::vector<Fr> canon(nnz);
      for (size_t k = 0; k < nnz; k++) canon[k] = c.val[k].from_mont();
      if ((e = ctx->staging(sizeof(Fr) * nnz)) != hipSuccess) return e;
      if ((e = hipMemcpy(ctx->d_tmp, canon.data(), sizeof(Fr) * nnz, hipMemcpyHostToDevice)) != hipSuccess) return e;
      if ((e = ntt_from_canonical(static_cast<const uint32_t*>(ctx->d_tmp), pk->d_val[m], (uint32_t)nnz, ctx->stream)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return e;
    }
  }
  if ((e = hipMalloc(&pk->a_query, sizeof(G1Affine) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->b_g1_query, sizeof(G1Affine) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->b_g2_query, sizeof(G2Affine) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->l_query, sizeof(G1Affine) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->h_query, sizeof(G1Affine) * N)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->a28, sizeof(Affine<Fq28>) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->b1_28, sizeof(Affine<Fq28>) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->b2_28, sizeof(Affine<Fq2_28>) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->l28, sizeof(Affine<Fq28>) * r->n_vars)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->h28, sizeof(Affine<Fq28>) * N)) != hipSuccess) return e;
  if ((e = hipMalloc(&pk->h28_rev, sizeof(Affine<Fq28>) * N)) != hipSuccess) return e;
  // group size: about 2^20 constraints in flight per group, at most 64 proofs (and at most 64 bucket partitions:
  // a proof above 2^16 constraints has several).  zkmi_ctx_set_group_size overrides (1 = never group).
  {
    uint32_t g = r->log_n <= 19 ? (1u << (r->log_n >= 14 ? 20 - r->log_n : 6)) : 1u;
    if (g > 64) g = 64;
    const uint32_t forced = ctx->group_override;  // zkmi_ctx_set_group_size
    if (forced >= 1 && forced <= 64 && (r->log_n <= 19 || forced == 1)) g = forced;
    const uint64_t capv = (1ull << r->log_n) > r->n_vars ? (1ull << r->log_n) : r->n_vars;
    const uint32_t parts = (uint32_t)msm_make_plan_shared(capv).nwin;
    while (g > 1 && g * parts > 64) g >>= 1;
    pk->gmax = g;
  }
  const uint64_t G = pk->gmax;
  for (int i = 0; i < zkmi_ctx::PROOF_RING; i++) {
    if ((e = hipMalloc(&pk->d_z[i], sizeof(Fr) * r->n_vars * G)) != hipSuccess) return e;
    if ((e = hipMalloc(&pk->d_h[i], 32ull * N * G)) != hipSuccess) return e;
  }
  if ((e = hipMalloc(&pk->d_zm, sizeof(Fr28) * r->n_vars * G)) != hipSuccess) return e;
  // a, b, c back to back in ONE buffer: the three inverse and the three forward transforms of the witness map run as
  // one batched launch per pass (3 x G vectors)
  // (a group of g <= G proofs uses the first 3 g N elements: a[0..g), b[0..g), c[0..g))
  if ((e = hipMalloc(&pk->d_a, sizeof(Fr28) * 3 * N * G)) != hipSuccess) return e;
  if ((e = hipHostMalloc(&pk->h_unsat, 2 * zkmi_ctx::PROOF_RING * sizeof(uint32_t), hipHostMallocCoherent)) != hipSuccess) return e;
  for (int i = 0; i < 2 * zkmi_ctx::PROOF_RING; i++) pk->h_unsat[i] = 0;
  pk->h_zent = pk->h_unsat + zkmi_ctx::PROOF_RING;
  const uint64_t cap = N > r->n_vars ? N : r->n_vars;
  if ((e = ctx->sort.reserve(cap, true)) != hipSuccess) return e;
  if (prover_sort_side() && (e = ctx->sort_z2.reserve(cap, true)) != hipSuccess) return e;
  if ((e = ctx->sort_h.reserve(cap, true)) != hipSuccess) return e;
  if ((e = ctx->g1.ensure_slots(MsmEngine<Fq28>::SLOTS)) != hipSuccess) return e;  // three proofs in flight x four G1 MSMs
  if ((e = ctx->g1.reserve(cap, true)) != hipSuccess) return e;
  if ((e = ctx->g2.reserve(cap, true)) != hipSuccess) return e;
  if (G > 1) {
    const MsmPlan sp = msm_make_plan_shared(cap);
    if ((e = ctx->sort.reserve_batch(cap, (uint32_t)G)) != hipSuccess) return e;
    if (prover_sort_side() && (e = ctx->sort_z2.reserve_batch(cap, (uint32_t)G)) != hipSuccess) return e;
    if ((e = ctx->sort_h.reserve_batch(cap, (uint32_t)G)) != hipSuccess) return e;
    if ((e = ctx->g1.reserve_buckets((uint64_t)sp.nb * sp.nwin * G)) != hipSuccess) return e;
    if ((e = ctx->g2.reserve_buckets((uint64_t)sp.nb * sp.nwin * G)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// delta_g1 / delta_g2 are final -> fixed-base tables of proof assembly
static void pk_build_delta_tables(zkmi_pk* pk) {
  pk->delta1_tab.reset(new FixedBase8<Fq>());
  pk->delta1_tab->build(pk->delta_g1);
  pk->delta2_tab.reset(new FixedBase8<Fq2>());
  pk->delta2_tab->build(pk->delta_g2);
  G1XYZZ k1 = G1XYZZ::from_affine(pk->beta_g1);
  k1.madd(pk->b1_0);
  pk->k1_tab.reset(new FixedBase8<Fq>());
  pk->k1_tab->build(k1.to_affine());
  G1XYZZ ka = G1XYZZ::from_affine(pk->alpha_g1);
  ka.madd(pk->a0);
  pk->ka = ka.to_affine();
  G2XYZZ kb = G2XYZZ::from_affine(pk->beta_g2);
  kb.madd(pk->b2_0);
  pk->kb2 = kb.to_affine();
}

// queries are final in the host representation -> build the device MSM copies
static hipError_t pk_convert_queries(zkmi_pk* pk) {
  hipStream_t st = pk->ctx->stream;
  const uint64_t N = 1ull << pk->log_n;
  hipError_t e;
  if ((e = bases_convert<Fq28>(pk->a_query, pk->a28, pk->n_vars, st)) != hipSuccess) return e;
  if ((e = bases_convert<Fq28>(pk->b_g1_query, pk->b1_28, pk->n_vars, st)) != hipSuccess) return e;
  if ((e = bases_convert<Fq2_28>(pk->b_g2_query, pk->b2_28, pk->n_vars, st)) != hipSuccess) return e;
  if ((e = bases_convert<Fq28>(pk->l_query, pk->l28, pk->n_vars, st)) != hipSuccess) return e;
  if ((e = bases_convert<Fq28>(pk->h_query, pk->h28, N, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_bitrev_points<Affine<Fq28>>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, pk->h28,
                     pk->h28_rev, (int)pk->log_n);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  // Precomputed tables 2^(c w) * P_i for every digit position (HBM is 288 GB: ~9 GB of
  // tables at N = 2^20): all digits of a scalar then share ONE bucket set, so the digit
  // width can grow to c = 20 (13 digits instead of 16 -> 19 % fewer bucket insertions)
  // and the host no longer walks a 255-doubling Horner chain.  (A/B library: ZKMI_MSM_SHARED=0 keeps the windowed schedule.)
  pk->shared = ZK_TUNE("ZKMI_MSM_SHARED", 1) != 0;
  if (!pk->shared) pk->gmax = 1;  // groups need the digit tables (their sort emits table indices)
  // B1's MSM is taken over r * z (prove_enqueue_z, "r B1 fold"): r * z per ring slot + a digit sort of its own
  // (A/B library: ZKMI_RB1_FOLD=0 none, 1 = one-proof groups only)
  const int fold = ZK_TUNE("ZKMI_RB1_FOLD", 2);
  if (pk->shared && (fold == 2 || (fold == 1 && pk->gmax == 1))) {
    zkmi_ctx* ctx = pk->ctx;
    const uint64_t cap = N > pk->n_vars ? N : pk->n_vars;
    for (int i = 0; i < zkmi_ctx::PROOF_RING; i++)
      if ((e = hipMalloc(&pk->d_rz[i], 32ull * pk->n_vars * pk->gmax)) != hipSuccess) return e;
    if ((e = ctx->sort_rz.reserve(cap, true)) != hipSuccess) return e;
    if (pk->gmax > 1 && (e = ctx->sort_rz.reserve_batch(cap, pk->gmax)) != hipSuccess) return e;
  }
  if (pk->shared) {
    const uint64_t nz = pk->n_vars - 1;
    const MsmPlan pz = msm_make_plan_shared(nz), ph = msm_make_plan_shared(N);
    if ((e = msm_build_table<Fq28>(pk->a28 + 1, nz, pz, &pk->a_tab, st)) != hipSuccess) return e;
    if ((e = msm_build_table<Fq28>(pk->b1_28 + 1, nz, pz, &pk->b1_tab, st)) != hipSuccess) return e;
    if ((e = msm_build_table<Fq28>(pk->l28 + 1, nz, pz, &pk->l_tab, st)) != hipSuccess) return e;
    if ((e = msm_build_table<Fq2_28>(pk->b2_28 + 1, nz, pz, &pk->b2_tab, st)) != hipSuccess) return e;
    if ((e = msm_build_table<Fq28>(pk->h28_rev, N, ph, &pk->h_tab, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  }
  return hipSuccess;
}

static Fr fr_from_u64(uint64_t v) {
  Fr a = Fr::zero();
  a.l[0] = (uint32_t)v;
  a.l[1] = (uint32_t)(v >> 32);
  return a.to_mont();
}

static void fr_limbs(const Fr& mont, uint32_t k[8]) {
  Fr c = mont.from_mont();
  memcpy(k, c.l, 32);
}

template <class F>
static hipError_t fixed_base_table(const Affine<F>& base, Affine<F>** d_table) {
  std::vector<Affine<F>> t(32 * 256);
  XYZZ<F> wbase = XYZZ<F>::from_affine(base);
  for (int w = 0; w < 32; w++) {
    XYZZ<F> acc = XYZZ<F>::infinity();
    t[w * 256] = Affine<F>::infinity();
    for (int d = 1; d < 256; d++) {
      acc.add(wbase);
      t[w * 256 + d] = acc.to_affine();
    }
    for (int k = 0; k < 8; k++) wbase.dbl_inplace();
  }
  hipError_t e = hipMalloc(d_table, sizeof(Affine<F>) * t.size());
  if (e != hipSuccess) return e;
  return hipMemcpy(*d_table, t.data(), sizeof(Affine<F>) * t.size(), hipMemcpyHostToDevice);
}

// out[i] = scalars[i] * base for Montgomery-form host scalars
template <class F>
static hipError_t fixed_base_batch(zkmi_ctx* ctx, const Affine<F>* d_table, const std::vector<Fr>& scalars_mont,
                                   Affine<F>* d_out) {
  const size_t n = scalars_mont.size();
  if (!n) return hipSuccess;
  std::vector<Fr> canon(n);
  for (size_t i = 0; i < n; i++) canon[i] = scalars_mont[i].from_mont();
  hipError_t e = ctx->staging(n * 32);
  if (e != hipSuccess) return e;
  if ((e = hipMemcpyAsync(ctx->d_tmp, canon.data(), n * 32, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_fixed_base<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream,
                     static_cast<const uint32_t*>(ctx->d_tmp), d_table, d_out, (uint32_t)n);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return hipStreamSynchronize(ctx->stream);
}

uint64_t zkmi_layout_pk() { return sizeof(zkmi_pk); }  // capi.hip zkmi_abi_layout_probe

extern "C" {

int32_t zkmi_groth16_setup(zkmi_ctx* ctx, const zkmi_r1cs* r, const uint8_t toxic[160], zkmi_pk** out_pk,
                           uint8_t* vk_out, uint64_t vk_cap) {
  ZK_ENTER(ctx);
  if (!ctx || !r || !toxic || !out_pk || !vk_out) return ZKMI_ERR_BAD_ARG;
  if (vk_cap < 672 + 96ull * r->n_pub) return ctx->fail(ZKMI_ERR_BAD_ARG, "vk buffer too small");
  Fr tau, alpha, beta, gamma, delta;
  Fr* tw[5] = {&tau, &alpha, &beta, &gamma, &delta};
  for (int i = 0; i < 5; i++)
    if (!fr_from_wire(toxic + 32 * i, tw[i])) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "toxic waste >= r");
  if (gamma.is_zero() || delta.is_zero()) return ctx->fail(ZKMI_ERR_BAD_ARG, "gamma/delta must be non-zero");
  const uint32_t N = 1u << r->log_n, nv = r->n_vars, np = r->n_pub, nc = r->n_constraints;
  // Lagrange basis at tau:  L_i = (tau^N - 1)/N * w^i / (tau - w^i)
  Fr tn = tau;
  for (uint32_t i = 0; i < r->log_n; i++) tn = tn.sqr();
  const Fr zt = tn - Fr::one();
  if (zt.is_zero()) return ctx->fail(ZKMI_ERR_BAD_ARG, "tau is in the evaluation domain");
  const Fr w = fr_root_of_unity((int)r->log_n);
  std::vector<Fr> L(N), den(N);
  {
    Fr wi = Fr::one();
    for (uint32_t i = 0; i < N; i++) {
      den[i] = tau - wi;
      L[i] = wi;
      wi = wi * w;
    }
    // batch inversion
    std::vector<Fr> pre(N);
    Fr run = Fr::one();
    for (uint32_t i = 0; i < N; i++) {
      pre[i] = run;
      run = run * den[i];
    }
    Fr inv = run.inv();
    const Fr scale = zt * fr_from_u64(N).inv();
    for (uint32_t i = N; i-- > 0;) {
      Fr di = inv * pre[i];
      inv = inv * den[i];
      L[i] = L[i] * di * scale;
    }
  }
  std::vector<Fr> a(nv, Fr::zero()), b(nv, Fr::zero()), c(nv, Fr::zero());
  for (uint32_t j = 0; j < np; j++) a[j] = L[nc + j];
  std::vector<Fr>* dst[3] = {&a, &b, &c};
  for (int m = 0; m < 3; m++) {
    const auto& M = r->m[m];
    for (uint32_t i = 0; i < nc; i++)
      for (uint32_t k = M.rowptr[i]; k < M.rowptr[i + 1]; k++)
        (*dst[m])[M.col[k]] = (*dst[m])[M.col[k]] + L[i] * M.val[k];
  }
  const Fr ginv = gamma.inv(), dinv = delta.inv();
  std::vector<Fr> lq(nv), hq(N);
  for (uint32_t j = 0; j < nv; j++) lq[j] = (beta * a[j] + alpha * b[j] + c[j]) * (j < np ? ginv : dinv);
  {
    Fr t = zt * dinv;
    for (uint32_t i = 0; i + 1 < N; i++) {
      hq[i] = t;
      t = t * tau;
    }
    hq[N - 1] = Fr::zero();
  }
  zkmi_pk* pk = new (std::nothrow) zkmi_pk();
  if (!pk) return ZKMI_ERR_BAD_ARG;
  hipError_t e = pk_alloc(pk, ctx, r);
  G1Affine* t1 = nullptr;
  G2Affine* t2 = nullptr;
  const G1Affine g1 = g1_generator();
  const G2Affine g2 = g2_generator();
  if (e == hipSuccess) e = fixed_base_table<Fq>(g1, &t1);
  if (e == hipSuccess) e = fixed_base_table<Fq2>(g2, &t2);
  if (e == hipSuccess) e = fixed_base_batch<Fq>(ctx, t1, a, pk->a_query);
  if (e == hipSuccess) e = fixed_base_batch<Fq>(ctx, t1, b, pk->b_g1_query);
  if (e == hipSuccess) e = fixed_base_batch<Fq2>(ctx, t2, b, pk->b_g2_query);
  if (e == hipSuccess) e = fixed_base_batch<Fq>(ctx, t1, lq, pk->l_query);
  if (e == hipSuccess) e = fixed_base_batch<Fq>(ctx, t1, hq, pk->h_query);
  std::vector<G1Affine> ic(np);
  if (e == hipSuccess) e = hipMemcpy(ic.data(), pk->l_query, sizeof(G1Affine) * np, hipMemcpyDeviceToHost);
  // the public part of the L query belongs to the verifying key; blank it in the proving key
  if (e == hipSuccess) e = hipMemset(pk->l_query, 0, sizeof(G1Affine) * np);
  if (e == hipSuccess) e = hipMemcpy(&pk->a0, pk->a_query, sizeof(G1Affine), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&pk->b1_0, pk->b_g1_query, sizeof(G1Affine), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&pk->b2_0, pk->b_g2_query, sizeof(G2Affine), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = pk_convert_queries(pk);
  if (t1) (void)hipFree(t1);
  if (t2) (void)hipFree(t2);
  if (e != hipSuccess) {
    delete pk;
    return ctx->hip_fail(e, "groth16 setup");
  }
  uint32_t k[8];
  auto mul1 = [&](const Fr& s) {
    fr_limbs(s, k);
    return scalar_mul(G1XYZZ::from_affine(g1), k, 8).to_affine();
  };
  auto mul2 = [&](const Fr& s) {
    fr_limbs(s, k);
    return scalar_mul(G2XYZZ::from_affine(g2), k, 8).to_affine();
  };
  pk->alpha_g1 = mul1(alpha);
  pk->beta_g1 = mul1(beta);
  pk->delta_g1 = mul1(delta);
  pk->beta_g2 = mul2(beta);
  pk->delta_g2 = mul2(delta);
  pk_build_delta_tables(pk);
  g1_to_wire(pk->alpha_g1, vk_out);
  g2_to_wire(pk->beta_g2, vk_out + 96);
  g2_to_wire(mul2(gamma), vk_out + 288);
  g2_to_wire(pk->delta_g2, vk_out + 480);
  for (uint32_t j = 0; j < np; j++) g1_to_wire(ic[j], vk_out + 672 + 96ull * j);
  *out_pk = pk;
  return ZKMI_OK;
}

int32_t zkmi_pk_load(zkmi_ctx* ctx, const zkmi_r1cs* r, const uint8_t alpha_g1[96], const uint8_t beta_g1[96],
                     const uint8_t beta_g2[192], const uint8_t delta_g1[96], const uint8_t delta_g2[192],
                     const uint8_t* a_query, const uint8_t* b_g1_query, const uint8_t* b_g2_query,
                     const uint8_t* h_query, const uint8_t* l_query, zkmi_pk** out_pk) {
  ZK_ENTER(ctx);
  if (!ctx || !r || !alpha_g1 || !beta_g1 || !beta_g2 || !delta_g1 || !delta_g2 || !a_query || !b_g1_query ||
      !b_g2_query || !h_query || !l_query || !out_pk)
    return ZKMI_ERR_BAD_ARG;
  zkmi_pk* pk = new (std::nothrow) zkmi_pk();
  if (!pk) return ZKMI_ERR_BAD_ARG;
  hipError_t e = pk_alloc(pk, ctx, r);
  if (e != hipSuccess) {
    delete pk;
    return ctx->hip_fail(e, "pk alloc");
  }
  const uint32_t N = 1u << r->log_n, nv = r->n_vars, np = r->n_pub;
  bool ok = g1_from_wire(alpha_g1, &pk->alpha_g1, true) && g1_from_wire(beta_g1, &pk->beta_g1, true) &&
            g1_from_wire(delta_g1, &pk->delta_g1, true) && g2_from_wire(beta_g2, &pk->beta_g2, true) &&
            g2_from_wire(delta_g2, &pk->delta_g2, true);
  std::vector<G1Affine> h1(nv > N ? nv : N);
  std::vector<G2Affine> h2(nv);
  auto up1 = [&](const uint8_t* src, uint32_t cnt, uint32_t pad_front, G1Affine* dst, uint32_t total) {
    for (uint32_t i = 0; i < total; i++) h1[i] = G1Affine::infinity();
    for (uint32_t i = 0; i < cnt && ok; i++) ok = g1_from_wire(src + 96ull * i, &h1[pad_front + i], true);
    if (ok) e = hipMemcpy(dst, h1.data(), sizeof(G1Affine) * total, hipMemcpyHostToDevice);
  };
  if (ok) up1(a_query, nv, 0, pk->a_query, nv);
  if (ok && e == hipSuccess) pk->a0 = h1[0];
  if (ok && e == hipSuccess) up1(b_g1_query, nv, 0, pk->b_g1_query, nv);
  if (ok && e == hipSuccess) pk->b1_0 = h1[0];
  if (ok && e == hipSuccess) up1(l_query, nv - np, np, pk->l_query, nv);
  if (ok && e == hipSuccess) up1(h_query, N - 1, 0, pk->h_query, N);
  for (uint32_t i = 0; i < nv && ok; i++) ok = g2_from_wire(b_g2_query + 192ull * i, &h2[i], true);
  if (ok && e == hipSuccess) {
    pk->b2_0 = h2[0];
    e = hipMemcpy(pk->b_g2_query, h2.data(), sizeof(G2Affine) * nv, hipMemcpyHostToDevice);
  }
  if (ok && e == hipSuccess) e = pk_convert_queries(pk);
  if (ok && e == hipSuccess) pk_build_delta_tables(pk);
  if (!ok || e != hipSuccess) {
    delete pk;
    return ok ? ctx->hip_fail(e, "pk upload") : ctx->fail(ZKMI_ERR_NON_CANONICAL, "proving key point invalid");
  }
  *out_pk = pk;
  return ZKMI_OK;
}

}  // extern "C"
uint32_t zkmi::pk_tree_height(const zkmi_pk* pk) { return pk ? pk->tree_height : 0; }
extern "C" {

int32_t zkmi_pk_free(zkmi_pk* pk) {
  if (!pk) return ZKMI_ERR_BAD_ARG;
  delete pk;
  return ZKMI_OK;
}

int32_t zkmi_pk_shape(const zkmi_pk* pk, uint32_t* n_vars, uint32_t* n_pub, uint32_t* log_n) {
  if (!pk) return ZKMI_ERR_BAD_ARG;
  if (n_vars) *n_vars = pk->n_vars;
  if (n_pub) *n_pub = pk->n_pub;
  if (log_n) *log_n = pk->log_n;
  return ZKMI_OK;
}

int32_t zkmi_pk_schedule_state(const zkmi_pk* pk, uint64_t out[3]) {
  if (!pk || !out) return ZKMI_ERR_BAD_ARG;
  out[0] = (pk->d_rz[0] != nullptr && pk->fold_dense) ? 1 : 0;
  out[1] = pk->last_entries;
  out[2] = pk->last_full;
  return ZKMI_OK;
}

int32_t zkmi_ctx_set_group_size(zkmi_ctx* ctx, uint32_t group) {
  if (!ctx || group > 64) return ZKMI_ERR_BAD_ARG;
  ctx->group_override = group;
  return ZKMI_OK;
}

int32_t zkmi_pk_export_g1_elems(const zkmi_pk* pk, uint8_t out_beta_g1[96], uint8_t out_delta_g1[96]) {
  if (!pk || !out_beta_g1 || !out_delta_g1) return ZKMI_ERR_BAD_ARG;
  g1_to_wire(pk->beta_g1, out_beta_g1);
  g1_to_wire(pk->delta_g1, out_delta_g1);
  return ZKMI_OK;
}

int32_t zkmi_pk_export_query(zkmi_ctx* ctx, const zkmi_pk* pk, int32_t which, uint64_t first, uint64_t count,
                             uint8_t* out) {
  ZK_ENTER(ctx);
  if (!ctx || !pk || !out || which < 0 || which > 4) return ZKMI_ERR_BAD_ARG;
  const uint64_t N = 1ull << pk->log_n;
  if (which == 2) {
    if (first + count > pk->n_vars) return ZKMI_ERR_BAD_ARG;
    std::vector<G2Affine> h(count);
    ZK_HIP(ctx, hipMemcpy(h.data(), pk->b_g2_query + first, sizeof(G2Affine) * count, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < count; i++) g2_to_wire(h[i], out + 192 * i);
    return ZKMI_OK;
  }
  const G1Affine* src = which == 0 ? pk->a_query : which == 1 ? pk->b_g1_query : which == 3 ? pk->h_query : pk->l_query + pk->n_pub;
  const uint64_t len = which == 3 ? N - 1 : which == 4 ? pk->n_vars - pk->n_pub : pk->n_vars;
  if (first + count > len) return ZKMI_ERR_BAD_ARG;
  std::vector<G1Affine> h(count);
  ZK_HIP(ctx, hipMemcpy(h.data(), src + first, sizeof(G1Affine) * count, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < count; i++) g1_to_wire(h[i], out + 96 * i);
  return ZKMI_OK;
}

// z -> h coefficients (Montgomery form, natural order) in pk->d_a
// The witness lands in pk->d_z on the main stream (the digit sort of the A/B/L MSMs reads it
// there); everything downstream of it (limb conversion, mat-vec, NTTs -> d_h) runs on `st`,
// which the prover points at its front stream so that it overlaps the z-MSMs.
// A group of G >= 1 witnesses (src[b]: host pointers if `host`, device pointers otherwise) goes through the map
// together: the vectors lie back to back in the ring buffers, every kernel takes the group as a batch dimension.
static int32_t witness_map_dev(zkmi_ctx* ctx, const zkmi_pk* pk, const void* const* src, bool host, uint32_t G,
                               hipStream_t st, int par = 0) {
  const uint32_t N = 1u << pk->log_n, nv = pk->n_vars;
  PhaseTimer* t = ctx->timer();
  // The copy runs on the copy stream: with a pinned host witness the upload of proof i+1 (32 B per variable
  // over PCIe) proceeds while the compute streams still work on proofs i-1 and i; the ring of witness
  // buffers makes that safe.  Canonicity (< r) is checked on the device, not in a host loop.
  pk->z_cur[par] = pk->d_z[par];
  bool aligned = true;  // 16-byte aligned device pointers (the vector loads of the sort and of k_gather_z)
  for (uint32_t b = 0; b < G; b++) aligned = aligned && (reinterpret_cast<uintptr_t>(src[b]) & 15u) == 0;
  if (host) {
    for (uint32_t b = 0; b < G; b++)
      ZK_HIP(ctx, hipMemcpyAsync(pk->d_z[par] + (size_t)b * nv, src[b], 32ull * nv, hipMemcpyHostToDevice, ctx->stream_copy));
  } else if (G == 1 && aligned) {
    pk->z_cur[par] = static_cast<const Fr*>(src[0]);  // read in place: the caller's buffer outlives the call
  } else if (G <= 64 && aligned) {
    ZPtrSet ps;
    for (uint32_t b = 0; b < 64; b++) ps.p[b] = static_cast<const uint4*>(src[b < G ? b : 0]);
    hipLaunchKernelGGL(k_gather_z, dim3((2 * nv + 63) / 64, G), dim3(64), 0, ctx->stream_copy, ps, reinterpret_cast<uint4*>(pk->d_z[par]), 2 * nv);
  } else {
    for (uint32_t b = 0; b < G; b++)
      ZK_HIP(ctx, hipMemcpyAsync(pk->d_z[par] + (size_t)b * nv, src[b], 32ull * nv, hipMemcpyDeviceToDevice, ctx->stream_copy));
  }
  ZK_HIP(ctx, hipEventRecord(ctx->ev_z[par], ctx->stream_copy));
  ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_z[par], 0));
  if (st != ctx->stream) ZK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_z[par], 0));
  if (t) t->begin(PH_WITNESS, st);
  hipLaunchKernelGGL(k_check_canonical, dim3((G * nv + 63) / 64), dim3(64), 0, st,
                     reinterpret_cast<const uint32_t*>(pk->z_cur[par]), G * nv, pk->h_unsat + par);
  ZK_HIP(ctx, ntt_from_canonical(reinterpret_cast<const uint32_t*>(pk->z_cur[par]), pk->d_zm, G * nv, st));
  MatSet ms;
  for (int m = 0; m < 3; m++) {
    ms.rowptr[m] = pk->d_rowptr[m];
    ms.col[m] = pk->d_col[m];
    ms.val[m] = pk->d_val[m];
  }
  if (G == 1 && pk->log_n <= 16)
    hipLaunchKernelGGL(k_matvec<8>, dim3((8 * N + 63) / 64, 1, 3), dim3(64), 0, st, ms, pk->d_zm, pk->d_a, pk->nc, N, pk->n_pub, nv);
  else
    hipLaunchKernelGGL(k_matvec<1>, dim3((N + 63) / 64, G, 3), dim3(64), 0, st, ms, pk->d_zm, pk->d_a, pk->nc, N, pk->n_pub, nv);
  if (pk->n_long_rows) hipLaunchKernelGGL(k_matvec_long, dim3(pk->n_long_rows, G), dim3(256), 0, st, ms, pk->d_long_rows, pk->d_zm, pk->d_a, N, nv);
  Fr28* const d_b = pk->d_a + (size_t)N * G;  // the layout follows the size of THIS group, not the key's maximum
  Fr28* const d_c = pk->d_a + 2 * (size_t)N * G;
  hipLaunchKernelGGL(k_check_sat, dim3((pk->nc + 64) / 64, G), dim3(64), 0, st, pk->d_a, d_b, d_c, pk->d_zm, pk->nc,
                     pk->h_unsat + par, N, nv);
  if (t) t->end(PH_WITNESS, st);
  hipError_t e;
  NttDomain* dom = ctx->domain((int)pk->log_n, &e);
  if (!dom) return ctx->hip_fail(e, "ntt domain");
  if (t) t->begin(PH_NTT, st);
  // evaluations -> coefficients (bit-reversed, scaled by g^i / N) -> evaluations on the coset
  // a, b, c together: two batched passes down, two up (3 x G vectors per launch instead of six launch pairs)
  const bool batch_abc = ZK_TUNE("ZKMI_WITNESS_BATCH", 1) != 0;  // A/B library: 0 = one transform per launch
  if (batch_abc) {
    ZK_HIP(ctx, dom->inverse_to_rev(pk->d_a, dom->rev_coset_n, nullptr, st, 3 * G));
    ZK_HIP(ctx, dom->forward_from_rev(pk->d_a, st, 3 * G));
  } else {
    for (int m = 0; m < 3; m++) {
      ZK_HIP(ctx, dom->inverse_to_rev(pk->d_a + (size_t)m * N * G, dom->rev_coset_n, nullptr, st, G));
      ZK_HIP(ctx, dom->forward_from_rev(pk->d_a + (size_t)m * N * G, st, G));
    }
  }
  // 1 / Z(g) with Z(g) = g^N - 1, g = 7
  Fr gn = fr_from_u64(7);
  for (uint32_t i = 0; i < pk->log_n; i++) gn = gn.sqr();
  const Fr zinv = (gn - Fr::one()).inv().from_mont();
  const Fr28 zinv28 = Fr28::from_canonical(zinv.l);
  hipLaunchKernelGGL(k_quotient, dim3((G * N + 63) / 64), dim3(64), 0, st, pk->d_a, d_b, d_c, zinv28, G * N);
  // h coefficients = coset iNTT, left in bit-reversed order as canonical words (H MSM digits)
  ZK_HIP(ctx, dom->inverse_to_rev(pk->d_a, dom->rev_coset_inv_n, pk->d_h[par], st, G));
  if (t) t->end(PH_NTT, st);
  if (st != ctx->stream) ZK_HIP(ctx, hipEventRecord(ctx->ev_h[par], st));
  ZK_HIP(ctx, hipGetLastError());
  return ZKMI_OK;
}

int32_t zkmi_groth16_witness_map(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* z, uint8_t* out_h) {
  ZK_ENTER(ctx);
  if (!ctx || !pk || !z || !out_h) return ZKMI_ERR_BAD_ARG;
  const void* src[1] = {z};
  int32_t rc = witness_map_dev(ctx, pk, src, true, 1, ctx->stream);
  if (rc != ZKMI_OK) return rc;
  const uint32_t N = 1u << pk->log_n;
  std::vector<uint8_t> rev(32ull * N);
  ZK_HIP(ctx, hipMemcpyAsync(rev.data(), pk->d_h[0], 32ull * N, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t flags0 = pk->h_unsat[0];
  pk->h_unsat[0] = 0;
  if (flags0 & 2u) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "witness element >= r");
  for (uint32_t p = 0; p < N; p++) {
    uint32_t i = 0;
    for (uint32_t b = 0; b < pk->log_n; b++) i |= ((p >> b) & 1u) << (pk->log_n - 1 - b);
    memcpy(out_h + 32ull * i, rev.data() + 32ull * p, 32);
  }
  return ZKMI_OK;
}

// Device work of one proof, queued on the ctx streams in two halves; MSM partials land in slot set
// `par` (ring of zkmi_ctx::PROOF_RING).  Nothing here blocks the host.
//   prove_enqueue_z : witness copy, witness map + NTTs (front stream), digit sort of z, the four MSMs over z
//   prove_enqueue_h : digit sort of h, the H MSM
// The batch prover queues proof i+1's first half BEFORE proof i's second half: the H MSM is the only
// consumer of the NTTs, and the NTT kernels (1024-thread blocks, 100 KB of LDS) only get workgroup
// slots in the gaps the accumulation kernels leave, so h arrives late; with the halves interleaved the
// main stream always has a full accumulation to run instead of waiting for it.
// solo = nothing else of this context is in flight or will be queued before the second half (zkmi_groth16_prove[_dev]): the
// second half's sort of h may then be queued by the first half
// followed = the first half of ANOTHER group will be queued between this group's two halves (every group of a batch but
// the last): only then do L and H share a bucket set (see below)
static int32_t prove_enqueue_z(zkmi_ctx* ctx, const zkmi_pk* pk, const void* const* src, bool host, uint32_t G, int par, bool solo = false,
                               bool followed = false, const uint8_t* r_bytes = nullptr) {
  const uint32_t nv = pk->n_vars;
  hipStream_t st = ctx->stream;
  // only the H MSM depends on the NTTs: the witness map runs on the front stream beside the MSMs over z
  int32_t rc = witness_map_dev(ctx, pk, src, host, G, ctx->stream_front, par);
  if (rc != ZKMI_OK) return rc;
  PhaseTimer* t = ctx->timer();
  const int s0 = 4 * par, g2s = par;
  // ZKMI_HEAVY_ON=1 (A/B library): heavy-bucket kernels on their own side stream (round 2; created on first use); default:
  // at the head of each MSM's reduction stream, 0: in line on the accumulation streams (msm_impl.hpp run_device)
  const bool heavy_side = ZK_TUNE("ZKMI_HEAVY_ON", 2) == 1;
  if (heavy_side) ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_heavy, false));
  const hipStream_t sth = heavy_side ? ctx->stream_heavy : nullptr;
  // MSMs over the assignment z[1..): one digit sort, four bucket passes.  Every
  // MSM's reduction runs on the aux stream behind its accumulation and leaves the
  // per-window partials in a pinned host slot + an event.
  const uint32_t* zs = reinterpret_cast<const uint32_t*>(pk->z_cur[par] + 1);
  const bool sh = pk->shared;
  // ZKMI_SORT_SIDE=1: the digit sort runs on its own (high-priority) stream into one of two buffer sets, so the sort
  // of proof i+1 overlaps the accumulations of proof i instead of standing between two accumulations on the main
  // stream.  Measured: no gain over sorting on the main stream once the reductions are spread over three streams.
  const bool sort_side = prover_sort_side();
  MsmSort& sz = (sort_side && (ctx->z_flip++ & 1u)) ? ctx->sort_z2 : ctx->sort;
  // the five reductions of a proof on three streams (A, L | B1, H | B2): one stream is ~90 % busy with their
  // latency chains and becomes the critical path (48.5 -> 50.0 proofs/s)
  const bool aux_split = prover_aux_split();
  const hipStream_t ra = ctx->stream_aux, rb = aux_split ? ctx->stream_aux2 : ctx->stream_aux,
                    rc2 = aux_split ? ctx->stream_aux3 : ctx->stream_aux;
  if (sort_side) ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_sort, true));
  const hipStream_t ss = sort_side ? ctx->stream_sort : st;
  if (sort_side) ZK_HIP(ctx, hipStreamWaitEvent(ss, ctx->ev_z[par], 0));  // the witness is in d_z
  if (G > 1)  // one digit sort for the whole group: bucket set b belongs to witness b (msm_sort.hip run_shared_batch)
    ZK_HIP(ctx, sz.run_shared_batch(zs, nv - 1, 8ull * nv, G, ss, t));
  else if (sh)
    ZK_HIP(ctx, sz.run_shared(zs, nv - 1, ss, t));
  else
    ZK_HIP(ctx, sz.run(zs, nv - 1, ss, t));
  if (sh && pk->d_rz[par]) hipLaunchKernelGGL(k_entries_to_host, dim3(1), dim3(64), 0, ss, sz.part_total, (uint32_t)sz.plan.nwin, pk->h_zent + par);
  // the G2 accumulation runs on its own stream beside the three G1 ones (same sort, disjoint
  // outputs): the kernels' drain tails overlap instead of adding up
  ZK_HIP(ctx, hipEventRecord(ctx->ev_sort[par], ss));
  if (sort_side) ZK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_sort[par], 0));
  ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_g2, ctx->ev_sort[par], 0));
  // one small proof cannot fill the chip with one accumulation (2^14 constraints: 256 waves for 1024 SIMDs): its
  // three G1 accumulations over z run side by side; anything bigger keeps them in line on the main stream
  const bool spread_on = ZK_TUNE("ZKMI_SPREAD", 1) != 0;
  const bool spread = spread_on && G == 1 && pk->log_n <= (solo ? prover_solo_max_log() : 16u) && !sort_side && sh;
  // (B1 borrows the copy stream, idle once the witness of this one proof is up; L gets a stream created on first use)
  // (a lone proof at the end of a batch; a proof by itself -- solo -- takes the fused launch below and needs neither)
  const bool three_streams = spread && !solo;
  if (three_streams) ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_acc3, false));
  const hipStream_t sb1 = three_streams ? ctx->stream_copy : st, sl = three_streams ? ctx->stream_acc3 : st;
  if (three_streams) {
    ZK_HIP(ctx, hipStreamWaitEvent(sb1, ctx->ev_sort[par], 0));
    ZK_HIP(ctx, hipStreamWaitEvent(sl, ctx->ev_sort[par], 0));
  }
  if (spread && solo) {
    // The latency of one small proof is a matter of placement: the transforms and the digit sorts are 1024-thread
    // workgroups, which find no CU while accumulation waves (168-256 registers each) sit on every SIMD -- in the trace the
    // last transform pass took 0.41 ms instead of 0.04 and the sort of h started 0.6 ms after h was ready.  So the sort
    // of h is queued here, on the front stream right behind the transforms, and the G1 accumulations wait for it: first
    // everything that needs whole CUs (0.5 ms), then the accumulations side by side.
    ZK_HIP(ctx, ctx->sort_h.run_shared(pk->d_h[par], 1u << pk->log_n, ctx->stream_front, t));
    ZK_HIP(ctx, hipEventRecord(ctx->ev_sorth[par], ctx->stream_front));
    // (G2 does not wait: its waves have no LDS and leave the sort's 1024-thread workgroups -- 8-42 registers -- room on
    // every SIMD; same box, ZKMI_SOLO_G2_EARLY=0 against the default: 2^14 2.45 vs 2.37 ms, 2^18 8.9 vs 8.25)
    const bool g2_early = ZK_TUNE("ZKMI_SOLO_G2_EARLY", 1) != 0;
    if (!g2_early) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_g2, ctx->ev_sorth[par], 0));
    ZK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_sorth[par], 0));
    // A, B1, L and H as ONE launch on the main stream, G2 beside it.  (On separate streams the grids did not start
    // together: a kernel that cannot place all its workgroups holds its dispatch pipe, and the streams sharing that pipe
    // wait -- L started when A had finished, H when the G2 heavy-bucket kernel had.)  The two sorts plan the same bucket
    // set unless n_vars is far below N; then H goes by itself (prove_enqueue_h).
    ZK_HIP(ctx, ctx->g2.run_device(sz, pk->b2_tab, ctx->stream_g2, rc2, t, PH_MSM_ACCUM_G2, PH_MSM_REDUCE_G2, g2s, sth));
    const MsmPlan &pz = sz.plan, &ph = ctx->sort_h.plan;
    const bool fuse_h = ZK_TUNE("ZKMI_SOLO_FUSE_H", 1) != 0;  // A/B library test switch: 0 = H by itself
    const bool fused = fuse_h && same_bucket_set(pz, ph);
    ctx->h_mode[par] = fused ? zkmi_ctx::H_FUSED : zkmi_ctx::H_SORTED;
    const MsmSort* sorts[4] = {&sz, &sz, &sz, &ctx->sort_h};
    const Affine<Fq28>* tabs[4] = {pk->a_tab, pk->b1_tab, pk->l_tab, pk->h_tab};
    // four reduction chains, four streams: the front and copy streams have nothing left to do for this proof
    const hipStream_t reds[4] = {ra, rb, ctx->stream_front, ctx->stream_copy};
    const int slots4[4] = {s0 + 0, s0 + 1, s0 + 2, s0 + 3};
    ZK_HIP(ctx, ctx->g1.run_device_multi(sorts, tabs, fused ? 4 : 3, st, reds, t, PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, slots4, sth));
    return ZKMI_OK;
  }
  // L and H only ever appear added together (C = s A + r B1 - rs delta + L + H): when the sorts of z and of h plan the same
  // bucket set, the L MSM stops after its redo pass and the H MSM's segment sums add L's bucket array to its own -- three
  // additions per bucket for the pair instead of four, one tree sum and one host combine instead of two (0.085 of 8.1 x 10^9
  // instructions of a 2^20 proof, which is bound by the instruction issue rate: DESIGN.md 4.1, 4.10).  The two
  // accumulations know nothing of each other, so one proof, the last group of a batch and groups of small proofs merge
  // like a pipelined 2^20 proof does.  A/B library, ZKMI_LH_MERGE: 0 = separate reductions; 1 = round 4's first form, the H
  // accumulation continuing INSIDE L's bucket array (k_accum_g1_nc<.., INTO>: its main stream has to wait for L's
  // heavy-bucket and redo kernels, so only one-proof groups that another group's first half follows take it).
  const int lh_mode = ZK_TUNE("ZKMI_LH_MERGE", 2);
  const uint32_t N = 1u << pk->log_n;
  const bool same_set = sh && same_bucket_set(sz.plan, G > 1 ? msm_make_plan_shared_batch(N, G) : msm_make_plan_shared(N));
  const bool nocall_g1 = ZK_TUNE("ZKMI_ACCUM", 3) == 2 || ZK_TUNE("ZKMI_ACCUM", 3) == 3;
  // Not for ONE proof by itself (zkmi_groth16_prove[_dev]): there the merged reduction is the tail of the proof -- three
  // additions per bucket instead of two after the last accumulation, where L's own reduction used to run beside the H
  // accumulation -- and its latency went from 18.9 to 20.1 ms at 2^20 for nothing (lh_merge_modes_ab.txt).
  const bool lh_merge = same_set && ((lh_mode == 2 && !solo) ||
                                     (lh_mode == 1 && nocall_g1 && followed && (G == 1 || ZK_TUNE("ZKMI_LH_MERGE_GROUPS", 0) == 1)));
  // The r B1 fold: B1 only enters the proof as r * B1 inside C, and r * MSM(b1, z) = MSM(b1, r z).  With a digit sort of
  // r z of its own (copy stream, beside the A and L accumulations) the B1 MSM becomes a third source of the L + H reduction:
  // four additions per bucket for the three instead of six, one tree sum and one host combine instead of three.  What it
  // costs: r z has full-width digits whatever z looked like (a witness of bits and small values sorts into the first
  // window only; r times it fills all 13), so B1 costs what an MSM over random scalars costs -- the price of one of five
  // MSMs at most, and nothing for a witness of hashes like the Shielder relations' (+2-3 % proofs/s at 2^20 and in the
  // groups of 2^14..2^18: profiles/r04/experiments/rb1_fold_ab.txt).  A/B library: ZKMI_RB1_FOLD=0 off, 1 one-proof groups only.
  const bool fold_b1 = lh_merge && lh_mode == 2 && r_bytes && pk->d_rz[par] != nullptr && (pk->fold_dense || ZK_TUNE("ZKMI_RB1_FOLD_ALWAYS", 0) == 1);
  ctx->h_mode[par] = fold_b1 ? zkmi_ctx::H_INTO_LB : lh_merge ? zkmi_ctx::H_INTO_L : zkmi_ctx::H_OWN;
  if (fold_b1) {
    RSet rs;
    for (uint32_t g = 0; g < G; g++) {
      uint32_t rw[8];
      memcpy(rw, r_bytes + 32ull * g, 32);
      rs.r[g] = Fr28::from_canonical(rw);
    }
    // on the copy stream, in order behind the upload / ev_z of this proof (A/B library, ZKMI_RB1_STREAM=1: a stream of its own)
    const bool own = ZK_TUNE("ZKMI_RB1_STREAM", 0) == 1;
    if (own) {
      ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_rz, false));
      ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_rz, ctx->ev_z[par], 0));
    }
    const hipStream_t sc = own ? ctx->stream_rz : ctx->stream_copy;
    hipLaunchKernelGGL(k_scale_canonical, dim3((nv - 1 + 63) / 64, G), dim3(64), 0, sc, zs, rs, pk->d_rz[par], nv - 1, 8ull * nv);
    if (G > 1)
      ZK_HIP(ctx, ctx->sort_rz.run_shared_batch(pk->d_rz[par], nv - 1, 8ull * nv, G, sc, t));
    else
      ZK_HIP(ctx, ctx->sort_rz.run_shared(pk->d_rz[par], nv - 1, sc, t));
    ZK_HIP(ctx, hipEventRecord(ctx->ev_rz[par], sc));
  }
  ZK_HIP(ctx, ctx->g2.run_device(sz, sh ? pk->b2_tab : pk->b2_28 + 1, ctx->stream_g2, rc2, t,
                                 PH_MSM_ACCUM_G2, PH_MSM_REDUCE_G2, g2s, sth));
  ZK_HIP(ctx, ctx->g1.run_device(sz, sh ? pk->a_tab : pk->a28 + 1, st, ra, t, PH_MSM_ACCUM_G1,
                                 PH_MSM_REDUCE_G1, s0 + 0, sth));
  // reductions: A and B1 on one stream, L (+ H) on the second, B2 on the third (without the merge: A, L | B1, H | B2)
  if (fold_b1) {
    ZK_HIP(ctx, hipStreamWaitEvent(sb1, ctx->ev_rz[par], 0));
    ZK_HIP(ctx, ctx->g1.run_device(ctx->sort_rz, pk->b1_tab, sb1, rb, t, PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, s0 + 1, sth, -1, MSM_RUN_NO_REDUCE));
  } else {
    ZK_HIP(ctx, ctx->g1.run_device(sz, sh ? pk->b1_tab : pk->b1_28 + 1, sb1, lh_merge ? ra : rb, t, PH_MSM_ACCUM_G1,
                                   PH_MSM_REDUCE_G1, s0 + 1, sth));
  }
  ZK_HIP(ctx, ctx->g1.run_device(sz, sh ? pk->l_tab : pk->l28 + 1, sl, lh_merge ? rb : ra, t, PH_MSM_ACCUM_G1,
                                 PH_MSM_REDUCE_G1, s0 + 2, sth, -1, lh_merge ? MSM_RUN_NO_REDUCE : 0));
  return ZKMI_OK;
}

static int32_t prove_enqueue_h(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t G, int par, bool solo = false) {
  const uint32_t N = 1u << pk->log_n;
  // one small proof (the solo case of prove_enqueue_z): h is already sorted (front stream, right behind the transforms) and
  // normally accumulated in the same launch as A, B1 and L; otherwise the H accumulation runs on the copy stream, idle by now
  (void)solo;  // what the first half decided is in ctx->h_mode[par]
  const int mode = ctx->h_mode[par];
  if (mode == zkmi_ctx::H_FUSED) return ZKMI_OK;  // accumulated with A, B1 and L
  const bool spread = mode == zkmi_ctx::H_SORTED;
  hipStream_t st = spread ? ctx->stream_copy : ctx->stream;
  if (spread) ZK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_sorth[par], 0));  // the sort of h, queued on the front stream
  PhaseTimer* t = ctx->timer();
  const bool sh = pk->shared;
  const bool heavy_side = ZK_TUNE("ZKMI_HEAVY_ON", 2) == 1;
  if (heavy_side) ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_heavy, false));
  const hipStream_t sth = heavy_side ? ctx->stream_heavy : nullptr;
  const bool sort_side = prover_sort_side();
  if (sort_side) ZK_HIP(ctx, ctx->lazy_stream(&ctx->stream_sort, true));
  const hipStream_t ss = sort_side ? ctx->stream_sort : st;
  ZK_HIP(ctx, hipStreamWaitEvent(ss, ctx->ev_h[par], 0));  // h coefficients from the front stream
  // H: all N coefficients (bit-reversed order) against the permuted h query; entry N-1 of the query is
  // infinity.  Own sort buffers (ctx->sort_h), written on the sort stream behind the previous H accumulation.
  if (spread)
    ;  // queued by prove_enqueue_z in front of the accumulations
  else if (G > 1)
    ZK_HIP(ctx, ctx->sort_h.run_shared_batch(pk->d_h[par], N, 8ull * N, G, ss, t));
  else if (sh)
    ZK_HIP(ctx, ctx->sort_h.run_shared(pk->d_h[par], N, ss, t));
  else
    ZK_HIP(ctx, ctx->sort_h.run(pk->d_h[par], N, ss, t));
  if (sort_side) {
    ZK_HIP(ctx, hipEventRecord(ctx->ev_sorth[par], ss));
    ZK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_sorth[par], 0));
  }
  const bool aux_split = prover_aux_split();
  const bool into_l = mode == zkmi_ctx::H_INTO_L || mode == zkmi_ctx::H_INTO_LB;
  if (into_l && (!same_bucket_set(ctx->sort_h.plan, ctx->g1.slot_plan[4 * par + 2]) ||
                 (mode == zkmi_ctx::H_INTO_LB && !same_bucket_set(ctx->sort_h.plan, ctx->g1.slot_plan[4 * par + 1]))))
    return ctx->fail(ZKMI_ERR_BAD_ARG, "internal: the sorts of z, r z and h planned different bucket sets");
  // (into_l: L's bucket array joins this MSM's segment sums -- or, A/B library with ZKMI_LH_MERGE=1, this MSM's kernels add
  // into L's array; the same reduction stream as the L accumulation's heavy-bucket and redo kernels)
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort_h, sh ? pk->h_tab : pk->h28_rev, st, aux_split ? ctx->stream_aux2 : ctx->stream_aux, t,
                                 PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, 4 * par + 3, sth, into_l ? 4 * par + 2 : -1,
                                 into_l && ZK_TUNE("ZKMI_LH_MERGE", 2) == 2 ? MSM_RUN_ADD_AT_REDUCE : 0,
                                 mode == zkmi_ctx::H_INTO_LB ? 4 * par + 1 : -1));
  return ZKMI_OK;
}

// Proof assembly, pure host arithmetic; the GPU may already be running the next proof.
//   A = alpha + a0 + MSM(a, z) + r delta            B = beta + b_0 + MSM(b, z) + s delta   (in G1 and in G2)
//   C = s A + r B_1 - r s delta + MSM(l, aux) + MSM(h, h)
// B_1 is not part of the proof; expanding it inside C, the r s delta terms cancel:
//   C = s A + r (beta_1 + b1_0) + r MSM(b1, z) + L + H
// so ONE multiplication by a variable point is left (s A; with r MSM(b1, z) when B1 was not folded into the device-side
// reduction: one doubling chain for both), and three by points the key fixes -- r delta_1, r (beta_1 + b1_0), s delta_2 --
// which come from 8-bit-window affine tables (32 mixed additions each).  Round 4 computed r delta, s delta, r s delta and
// s delta_2 from 4-bit tables of XYZZ points and always ran the two-point chain: 0.51 ms of one CPU per proof against ~0.25
// now -- what a rank's share of the node's CPUs is spent on when proofs of 2^14 constraints leave at 2 700 per second.
struct AssemblyHead {
  G1XYZZ g_a, g_c;
  G2XYZZ g2_b;
};
// What assembly can compute from (r, s) and the key alone, i.e. before any MSM has finished (a single proof computes them
// while the GPU works)
struct AssemblyPre {
  uint32_t rk[8], sk[8];
  G1XYZZ d_r, r_k1;  // r * delta_1, r * (beta_1 + b1_0)
  G2XYZZ d2_s;       // s * delta_2
};
static AssemblyPre assemble_pre(const zkmi_pk* pk, const uint8_t r_bytes[32], const uint8_t s_bytes[32]) {
  AssemblyPre p;
  memcpy(p.rk, r_bytes, 32);
  memcpy(p.sk, s_bytes, 32);
  p.d_r = pk->delta1_tab->mul(p.rk);
  p.r_k1 = pk->k1_tab->mul(p.rk);
  p.d2_s = pk->delta2_tab->mul(p.sk);
  return p;
}
// A and s * A + r * B1 - rs * delta once the A and B1 MSMs are in (acc_b1 = infinity: B1's MSM reaches C through the
// device-side reduction it shares with L and H)
static void assemble_g1(const zkmi_pk* pk, const AssemblyPre& p, const G1XYZZ& acc_a, const G1XYZZ& acc_b1, AssemblyHead& h) {
  h.g_a = p.d_r;
  h.g_a.madd(pk->ka);
  h.g_a.add(acc_a);
  h.g_c = acc_b1.is_inf() ? scalar_mul_w4(h.g_a, p.sk) : scalar_mul2(h.g_a, p.sk, acc_b1, p.rk);
  h.g_c.add(p.r_k1);
}
// B once the G2 MSM is in: two additions
static void assemble_g2(const zkmi_pk* pk, const AssemblyPre& p, const G2XYZZ& acc_b2, AssemblyHead& h) {
  h.g2_b = p.d2_s;
  h.g2_b.madd(pk->kb2);
  h.g2_b.add(acc_b2);
}
static AssemblyHead assemble_head(const zkmi_pk* pk, const G1XYZZ& acc_a, const G1XYZZ& acc_b1, const G2XYZZ& acc_b2,
                                  const uint8_t r_bytes[32], const uint8_t s_bytes[32]) {
  const AssemblyPre p = assemble_pre(pk, r_bytes, s_bytes);
  AssemblyHead h;
  assemble_g1(pk, p, acc_a, acc_b1, h);
  assemble_g2(pk, p, acc_b2, h);
  return h;
}
// Stage 2: C += L + H, compression of A, B, C
static void assemble_tail(AssemblyHead& h, const G1XYZZ& acc_l, const G1XYZZ& acc_h, uint8_t out_proof[192]) {
  h.g_c.add(acc_l);
  h.g_c.add(acc_h);
  g1_compress(h.g_a.to_affine(), out_proof);
  g2_compress(h.g2_b.to_affine(), out_proof + 48);
  g1_compress(h.g_c.to_affine(), out_proof + 144);
}

// Host part: wait for the slot set's partials, combine windows, assemble and compress the G proofs of the group
// (on several host threads when G > 1).  The GPU may already be running the next group.
static int32_t prove_finish(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* r_bytes, const uint8_t* s_bytes, uint32_t G,
                            int par, uint8_t* out_proofs) {
  const int s0 = 4 * par, g2s = par;
  // called once the A MSM of this slot is in (its accumulation ran behind the digit sort of z and k_entries_to_host)
  auto note_density = [&]() {
    if (!pk->d_rz[par]) return;
    // digit positions that can be non-zero: the top one is empty when the digit width divides 255 (msm_make_plan_shared)
    const MsmPlan& zp = ctx->g1.slot_plan[s0];
    const int live = zp.ndigits - ((zp.ndigits - 1) * zp.c >= 255 ? 1 : 0);
    const uint64_t full = (uint64_t)(pk->n_vars - 1) * G * (uint64_t)live;
    pk->last_entries = pk->h_zent[par];
    pk->last_full = full;
    pk->fold_dense = 10ull * pk->h_zent[par] >= 9ull * full;
  };
  const bool merged_b1 = ctx->h_mode[par] == zkmi_ctx::H_INTO_LB;             // acc_h arrives as r * B1 + L + H
  const bool merged = ctx->h_mode[par] == zkmi_ctx::H_INTO_L || merged_b1;  // acc_h arrives as L + H
  std::vector<G1XYZZ> acc_a(G), acc_b1(G), acc_l(G), acc_h(G);
  std::vector<G2XYZZ> acc_b2(G);
  if (G == 1) {
    const bool lat_debug = debug_level() >= 2;
    const auto t0 = std::chrono::steady_clock::now();
    long tm[6] = {0, 0, 0, 0, 0, 0};
    auto mark = [&](int i) { tm[i] = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
    // one proof: the host's scalar multiplications run while the GPU works -- the fixed-base ones before anything is
    // waited for, s * A + r * B1 as soon as the two G1 MSMs over z are in (the G2 MSM takes three times as long)
    AssemblyHead head;
    const AssemblyPre pre = assemble_pre(pk, r_bytes, s_bytes);
    if (ctx->g1.host_spin && !merged && ZK_TUNE("ZKMI_SOLO_EVENT_ORDER", 1) != 0) {
      // ONE proof by itself: the five results are taken in the order they arrive.  A small proof's G2 MSM lands first (one
      // reduction chain against four) and B1 before A: B is assembled and compressed, r * B1 and s * A are multiplied out
      // while the remaining reductions still run, and what is left behind the last result is three additions and one shared
      // inversion.  (In the fixed order A, B1 | s A + r B1 | B2 | L | H | tail the host worked for 0.40 ms AFTER A had landed
      // at 2^14: 0.19 ms of the joint doubling chain, 0.12 ms of combining and assembling a B2 that had been ready for
      // 0.3 ms, 0.07 ms of three inversions.)
      enum { RA = 0, RB1 = 1, RL = 2, RH = 3, RB2 = 4 };
      hipEvent_t evs[5] = {ctx->g1.done[s0 + 0], ctx->g1.done[s0 + 1], ctx->g1.done[s0 + 2], ctx->g1.done[s0 + 3], ctx->g2.done[g2s]};
      bool got[5] = {false, false, false, false, false};
      G1XYZZ s_a = G1XYZZ::infinity(), r_b1 = G1XYZZ::infinity();
      uint8_t b_bytes[96];  // compressed B: copied out behind the witness checks only (a rejected witness leaves nothing in out_proofs)
      int left = 5;
      bool polled_busy = false;
      while (left) {
        bool progress = false;
        for (int i = 0; i < 5; i++) {
          if (got[i]) continue;
          const hipError_t q = hipEventQuery(evs[i]);
          if (q == hipErrorNotReady) {
            polled_busy = true;
            continue;
          }
          if (q != hipSuccess) return ctx->hip_fail(q, "hipEventQuery(done[slot])");
          got[i] = true;
          left--;
          progress = true;
          switch (i) {
            case RA:
              ZK_HIP(ctx, ctx->g1.finish_host(&acc_a[0], s0 + 0));
              note_density();
              head.g_a = pre.d_r;
              head.g_a.madd(pk->ka);
              head.g_a.add(acc_a[0]);
              s_a = scalar_mul_w4(head.g_a, pre.sk);
              break;
            case RB1:
              ZK_HIP(ctx, ctx->g1.finish_host(&acc_b1[0], s0 + 1));
              r_b1 = scalar_mul_w4(acc_b1[0], pre.rk);
              break;
            case RL: ZK_HIP(ctx, ctx->g1.finish_host(&acc_l[0], s0 + 2)); break;
            case RH: ZK_HIP(ctx, ctx->g1.finish_host(&acc_h[0], s0 + 3)); break;
            case RB2:
              ZK_HIP(ctx, ctx->g2.finish_host(&acc_b2[0], g2s));
              assemble_g2(pk, pre, acc_b2[0], head);
              g2_compress(head.g2_b.to_affine(), b_bytes);
              break;
          }
          if (lat_debug) mark(i == RB2 ? 5 : i);
          break;  // after the work, look at all pending results again
        }
        (void)progress;
      }
      if (polled_busy) (void)hipGetLastError();  // hipErrorNotReady is a status, not a failure (host_pool.hpp wait_event)
      const uint32_t flags = pk->h_unsat[par];
      pk->h_unsat[par] = 0;
      if (flags & 2u) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "witness element >= r");
      if (flags) return ctx->fail(ZKMI_ERR_UNSATISFIED, "assignment does not satisfy the relation (or z[0] != 1)");
      head.g_c = s_a;
      head.g_c.add(r_b1);
      head.g_c.add(pre.r_k1);
      head.g_c.add(acc_l[0]);
      head.g_c.add(acc_h[0]);
      const G1XYZZ ac[2] = {head.g_a, head.g_c};
      G1Affine aff[2];
      batch_to_affine(ac, 2, aff);
      g1_compress(aff[0], out_proofs);
      memcpy(out_proofs + 48, b_bytes, 96);
      g1_compress(aff[1], out_proofs + 144);
      if (lat_debug) {
        const long t_end = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "zkmi: finish (arrival order): A in + s A done %ld us, B1 + r B1 %ld, L %ld, H %ld, B2 + B compressed %ld, proof out %ld\n", tm[0], tm[1], tm[2],
                tm[3], tm[5], t_end);
      }
      return ZKMI_OK;
    }
    ZK_HIP(ctx, ctx->g1.finish_host(&acc_a[0], s0 + 0));
    note_density();
    if (merged_b1) acc_b1[0] = G1XYZZ::infinity();  // r * MSM(b1, z) comes in through the H slot: assemble_g1 multiplies the fixed part of B1 only
    else ZK_HIP(ctx, ctx->g1.finish_host(&acc_b1[0], s0 + 1));
    mark(0);
    assemble_g1(pk, pre, acc_a[0], acc_b1[0], head);
    mark(1);
    ZK_HIP(ctx, ctx->g2.finish_host(&acc_b2[0], g2s));
    assemble_g2(pk, pre, acc_b2[0], head);
    mark(2);
    if (merged) acc_l[0] = G1XYZZ::infinity();  // its buckets went into the H MSM's reduction
    else ZK_HIP(ctx, ctx->g1.finish_host(&acc_l[0], s0 + 2));
    mark(3);
    ZK_HIP(ctx, ctx->g1.finish_host(&acc_h[0], s0 + 3));
    mark(4);
    const uint32_t flags = pk->h_unsat[par];
    pk->h_unsat[par] = 0;  // (the slot's next proof is queued after this function returns)
    if (flags & 2u) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "witness element >= r");
    if (flags) return ctx->fail(ZKMI_ERR_UNSATISFIED, "assignment does not satisfy the relation (or z[0] != 1)");
    assemble_tail(head, acc_l[0], acc_h[0], out_proofs);
    mark(5);
    if (lat_debug)
      fprintf(stderr, "zkmi: finish: A,B1 ready %ld us, G1 part assembled %ld, B2 ready + B assembled %ld, L ready %ld, H ready %ld, tail done %ld\n", tm[0], tm[1],
              tm[2], tm[3], tm[4], tm[5]);
    return ZKMI_OK;
  }
  // A group: the driving thread only waits for the slots' events; every proof's share of the work -- combining its
  // partition sums into the five MSM results (~1 400 field products), the scalar multiplications of assembly -- is one task
  // of the process's persistent pool, within the CPUs this rank may use (host_pool.hpp).
  ZK_HIP(ctx, ctx->g1.wait_slot(s0 + 0));
  note_density();
  if (!merged_b1) ZK_HIP(ctx, ctx->g1.wait_slot(s0 + 1));
  if (!merged) ZK_HIP(ctx, ctx->g1.wait_slot(s0 + 2));
  ZK_HIP(ctx, ctx->g2.wait_slot(g2s));
  ZK_HIP(ctx, ctx->g1.wait_slot(s0 + 3));
  // the flag words precede the h coefficients on the front stream, which the H MSM waited for
  {
    const uint32_t flags = pk->h_unsat[par];
    pk->h_unsat[par] = 0;
    if (flags & 2u) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "witness element >= r");
    if (flags) return ctx->fail(ZKMI_ERR_UNSATISFIED, "assignment does not satisfy the relation (or z[0] != 1)");
  }
  std::vector<AssemblyHead> heads(G);
  const std::function<void(uint32_t)> one = [&](uint32_t b) {
    const G1XYZZ inf = G1XYZZ::infinity();
    const G1XYZZ a = ctx->g1.host_result_vec(s0 + 0, (int)b);
    const G1XYZZ b1 = merged_b1 ? inf : ctx->g1.host_result_vec(s0 + 1, (int)b);  // folded: arrives inside the H slot's sum
    const G2XYZZ b2 = ctx->g2.host_result_vec(g2s, (int)b);
    AssemblyHead h = assemble_head(pk, a, b1, b2, r_bytes + 32ull * b, s_bytes + 32ull * b);
    if (!merged) h.g_c.add(ctx->g1.host_result_vec(s0 + 2, (int)b));
    h.g_c.add(ctx->g1.host_result_vec(s0 + 3, (int)b));
    heads[b] = h;
  };
  HostPool::instance().run(G, host_cpu_budget(), one);
  // the 3 G proof elements of the group are normalised with ONE base-field inversion per curve instead of one each
  std::vector<G1XYZZ> p1(2 * (size_t)G);
  std::vector<G2XYZZ> p2(G);
  for (uint32_t b = 0; b < G; b++) {
    p1[2 * b] = heads[b].g_a;
    p1[2 * b + 1] = heads[b].g_c;
    p2[b] = heads[b].g2_b;
  }
  std::vector<G1Affine> a1(p1.size());
  std::vector<G2Affine> a2(p2.size());
  batch_to_affine(p1.data(), p1.size(), a1.data());
  batch_to_affine(p2.data(), p2.size(), a2.data());
  for (uint32_t b = 0; b < G; b++) {
    uint8_t* out = out_proofs + 192ull * b;
    g1_compress(a1[2 * b], out);
    g2_compress(a2[b], out + 48);
    g1_compress(a1[2 * b + 1], out + 144);
  }
  return ZKMI_OK;
}

static int32_t prove_impl(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* z, const void* d_z, const uint8_t r_bytes[32],
                          const uint8_t s_bytes[32], uint8_t out_proof[192]) {
  if (!ctx || !pk || (!z && !d_z) || !r_bytes || !s_bytes || !out_proof) return ZKMI_ERR_BAD_ARG;
  if (!fr_is_canonical(r_bytes) || !fr_is_canonical(s_bytes)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "r/s >= r");
  const void* src[1] = {z ? static_cast<const void*>(z) : d_z};
  // ZKMI_DEBUG=2: host-side timestamps of one proof (queueing the two halves, waiting + assembly) on stderr
  const bool lat_debug = debug_level() >= 2;
  const auto t0 = std::chrono::steady_clock::now();
  auto us = [&] { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
  int32_t rc = prove_enqueue_z(ctx, pk, src, z != nullptr, 1, 0, true, false, r_bytes);
  const long t_z = us();
  if (rc == ZKMI_OK) rc = prove_enqueue_h(ctx, pk, 1, 0, true);
  const long t_h = us();
  if (rc == ZKMI_OK) rc = prove_finish(ctx, pk, r_bytes, s_bytes, 1, 0, out_proof);
  if (lat_debug) fprintf(stderr, "zkmi: single proof 2^%u: queued z-half %ld us, h-half %ld us, finished %ld us\n", pk->log_n, t_z, t_h, us());
  if (rc != ZKMI_OK) {
    // whatever was queued before the failure still reads the caller's witness and the key: wait for it
    const std::string msg = ctx->err;
    (void)ctx->drain();
    (void)ctx->g1.reset_transients();  // an MSM abandoned between its accumulation and its redo pass leaves a list behind
    (void)ctx->g2.reset_transients();
    for (int i = 0; i < zkmi_ctx::PROOF_RING; i++) pk->h_unsat[i] = 0;  // flags of proofs that were queued and never finished
    ctx->err = msg;
  }
  return rc;
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
// Host self-test of the scalar multiplications of proof assembly (curve.hpp FixedBase4, scalar_mul2) against the plain
// double-and-add, in G1 and G2, on seeded scalars including 0, 1 and r - 1.  No GPU involved.
int32_t zkmi_selftest_assembly(uint64_t seed, uint32_t iters, uint32_t* out_mismatches) {
  if (!out_mismatches) return ZKMI_ERR_BAD_ARG;
  uint64_t st = seed;
  auto next = [&]() {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto scalar = [&](uint32_t it, uint32_t k[8]) {
    for (int i = 0; i < 8; i += 2) {
      const uint64_t v = next();
      k[i] = (uint32_t)v;
      k[i + 1] = (uint32_t)(v >> 32);
    }
    k[7] &= 0x3fffffffu;  // < 2^254 < r
    if (it % 5 == 1) memset(k, 0, 32);
    if (it % 5 == 2) {
      memset(k, 0, 32);
      k[0] = 1;
    }
    if (it % 5 == 3) {
      memcpy(k, FrParams::MOD, 32);
      k[0] -= 1;  // r - 1
    }
  };
  auto same1 = [](const G1XYZZ& a, const G1XYZZ& b) {
    const G1Affine x = a.to_affine(), y = b.to_affine();
    return x.x == y.x && x.y == y.y;
  };
  auto same2 = [](const G2XYZZ& a, const G2XYZZ& b) {
    const G2Affine x = a.to_affine(), y = b.to_affine();
    return x.x == y.x && x.y == y.y;
  };
  uint32_t bad = 0;
  uint32_t k[8], k2[8];
  scalar(0, k);
  const G1Affine p1 = scalar_mul(G1XYZZ::from_affine(g1_generator()), k, 8).to_affine();
  scalar(0, k);
  const G2Affine p2 = scalar_mul(G2XYZZ::from_affine(g2_generator()), k, 8).to_affine();
  std::unique_ptr<FixedBase4<Fq>> t1(new FixedBase4<Fq>());
  std::unique_ptr<FixedBase4<Fq2>> t2(new FixedBase4<Fq2>());
  t1->build(p1);
  t2->build(p2);
  // what assembly uses since round 5: 8-bit windows over affine entries, the one-point window multiplication, the shared inversion
  std::unique_ptr<FixedBase8<Fq>> u1(new FixedBase8<Fq>());
  std::unique_ptr<FixedBase8<Fq2>> u2(new FixedBase8<Fq2>());
  u1->build(p1);
  u2->build(p2);
  std::vector<G1XYZZ> n1;
  std::vector<G2XYZZ> n2;
  const G1XYZZ q1 = scalar_mul(G1XYZZ::from_affine(p1), k, 8);  // a second, unrelated G1 point
  for (uint32_t it = 0; it < iters; it++) {
    scalar(it, k);
    scalar(it + 2, k2);
    if (!same1(t1->mul(k), scalar_mul(G1XYZZ::from_affine(p1), k, 8))) bad++;
    if (!same2(t2->mul(k), scalar_mul(G2XYZZ::from_affine(p2), k, 8))) bad++;
    if (!same1(u1->mul(k), scalar_mul(G1XYZZ::from_affine(p1), k, 8))) bad++;
    if (!same2(u2->mul(k), scalar_mul(G2XYZZ::from_affine(p2), k, 8))) bad++;
    if (!same1(scalar_mul_w4(q1, k), scalar_mul(q1, k, 8))) bad++;
    n1.push_back(u1->mul(k));  // (infinity for k = 0: the batch must carry it through)
    n2.push_back(u2->mul(k2));
    G1XYZZ want = scalar_mul(G1XYZZ::from_affine(p1), k, 8);
    want.add(scalar_mul(q1, k2, 8));
    if (!same1(scalar_mul2(G1XYZZ::from_affine(p1), k, q1, k2), want)) bad++;
    // equal and opposite points: the shared doubling chain must survive P + P and P - P
    G1XYZZ dbl = scalar_mul(G1XYZZ::from_affine(p1), k, 8);
    dbl.add(scalar_mul(G1XYZZ::from_affine(p1), k2, 8));
    if (!same1(scalar_mul2(G1XYZZ::from_affine(p1), k, G1XYZZ::from_affine(p1), k2), dbl)) bad++;
    G1XYZZ opp = scalar_mul(G1XYZZ::from_affine(p1), k, 8);
    opp.add(scalar_mul(G1XYZZ::from_affine(p1).neg(), k, 8));
    if (!opp.is_inf() || !scalar_mul2(G1XYZZ::from_affine(p1), k, G1XYZZ::from_affine(p1).neg(), k).is_inf()) bad++;
  }
  {
    std::vector<G1Affine> a1(n1.size());
    std::vector<G2Affine> a2(n2.size());
    batch_to_affine(n1.data(), n1.size(), a1.data());
    batch_to_affine(n2.data(), n2.size(), a2.data());
    for (size_t i = 0; i < n1.size(); i++) {
      const G1Affine w = n1[i].to_affine();
      if (!(a1[i].x == w.x && a1[i].y == w.y)) bad++;
    }
    for (size_t i = 0; i < n2.size(); i++) {
      const G2Affine w = n2[i].to_affine();
      if (!(a2[i].x == w.x && a2[i].y == w.y)) bad++;
    }
  }
  *out_mismatches = bad;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

int32_t zkmi_groth16_prove(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* z, const uint8_t r_bytes[32],
                           const uint8_t s_bytes[32], uint8_t out_proof[192]) {
  ZK_ENTER(ctx);
  if (!z) return ZKMI_ERR_BAD_ARG;
  return prove_impl(ctx, pk, z, nullptr, r_bytes, s_bytes, out_proof);
}

int32_t zkmi_groth16_prove_dev(zkmi_ctx* ctx, const zkmi_pk* pk, const void* d_z, const uint8_t r_bytes[32],
                               const uint8_t s_bytes[32], uint8_t out_proof[192]) {
  ZK_ENTER(ctx);
  if (!d_z) return ZKMI_ERR_BAD_ARG;
  return prove_impl(ctx, pk, nullptr, d_z, r_bytes, s_bytes, out_proof);
}

// Batch of independent proofs over one key (BASELINE config 2), up to three in flight: the device queue
// always holds proof i+1's z-half and proof i's h-half while the CPU assembles proof i-1.
//   main stream:  z(0) | z(1) h(0) | z(2) h(1) | ...      host:  finish(0) after queueing [z(2) h(1)], ...
static int32_t prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t n_proofs, const void* const* d_z, bool host,
                           const uint8_t* r_bytes, const uint8_t* s_bytes, uint8_t* out_proofs) {
  if (!ctx || !pk || !d_z || !r_bytes || !s_bytes || !out_proofs) return ZKMI_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n_proofs; i++) {
    if (!d_z[i]) return ZKMI_ERR_BAD_ARG;
    if (!fr_is_canonical(r_bytes + 32ull * i) || !fr_is_canonical(s_bytes + 32ull * i))
      return ctx->fail(ZKMI_ERR_NON_CANONICAL, "r/s >= r");
  }
  if (n_proofs == 0) return ZKMI_OK;
  constexpr int RING = zkmi_ctx::PROOF_RING;
  // the driving thread of a batch sleeps between polls while it waits for the GPU (host_pool.hpp); restored on every exit
  struct SpinGuard {
    zkmi_ctx* c;
    explicit SpinGuard(zkmi_ctx* x) : c(x) { c->g1.host_spin = c->g2.host_spin = false; }
    ~SpinGuard() { c->g1.host_spin = c->g2.host_spin = true; }
  } spin_guard(ctx);
  // on an error the other proofs in flight still have work queued on every ctx stream: drain them
  // before handing control (and the right to free buffers) back to the caller
  auto bail = [&](int32_t code) {
    const std::string msg = ctx->err;
    (void)ctx->drain();
    (void)ctx->g1.reset_transients();
    (void)ctx->g2.reset_transients();
    for (int i = 0; i < zkmi_ctx::PROOF_RING; i++) pk->h_unsat[i] = 0;
    ctx->err = msg;
    return code;
  };
  // units of the pipeline = groups of up to pk->gmax proofs (1 above 2^16 constraints)
  const uint32_t gsz = pk->gmax;
  const uint32_t n_groups = (n_proofs + gsz - 1) / gsz;
  auto first = [&](uint32_t g) { return g * gsz; };
  auto count = [&](uint32_t g) { return (first(g) + gsz <= n_proofs) ? gsz : n_proofs - first(g); };
  auto finish = [&](uint32_t g) {
    return prove_finish(ctx, pk, r_bytes + 32ull * first(g), s_bytes + 32ull * first(g), count(g), (int)(g % RING),
                        out_proofs + 192ull * first(g));
  };
  auto enqueue_z = [&](uint32_t g) {
    return prove_enqueue_z(ctx, pk, d_z + first(g), host, count(g), (int)(g % RING), false, g + 1 < n_groups, r_bytes + 32ull * first(g));
  };
  int32_t rc = enqueue_z(0);
  if (rc != ZKMI_OK) return bail(rc);
  for (uint32_t g = 0; g < n_groups; g++) {
    if (g + 1 < n_groups) {
      rc = enqueue_z(g + 1);
      if (rc != ZKMI_OK) return bail(rc);
    }
    rc = prove_enqueue_h(ctx, pk, count(g), (int)(g % RING));
    if (rc != ZKMI_OK) return bail(rc);
    if (g >= 1 && (rc = finish(g - 1)) != ZKMI_OK) return bail(rc);  // frees slot set (g - 1) % RING = (g + 2) % RING
  }
  if ((rc = finish(n_groups - 1)) != ZKMI_OK) return bail(rc);
  return ZKMI_OK;
}

int32_t zkmi_groth16_prove_batch_dev(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t n_proofs, const void* const* d_z,
                                     const uint8_t* r_bytes, const uint8_t* s_bytes, uint8_t* out_proofs) {
  ZK_ENTER(ctx);
  return prove_batch(ctx, pk, n_proofs, d_z, false, r_bytes, s_bytes, out_proofs);
}

// The same pipeline fed from HOST witnesses: each upload runs on the copy stream while earlier proofs
// compute (pin the buffers -- hipHostMalloc / hipHostRegister / torch pin_memory -- or the runtime stages
// the copy synchronously).  This is the PCIe-inclusive rate bench.py reports as value_incl_h2d.
int32_t zkmi_groth16_prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t n_proofs, const uint8_t* const* z,
                                 const uint8_t* r_bytes, const uint8_t* s_bytes, uint8_t* out_proofs) {
  ZK_ENTER(ctx);
  return prove_batch(ctx, pk, n_proofs, reinterpret_cast<const void* const*>(z), true, r_bytes, s_bytes, out_proofs);
}

// One batch over SEVERAL GPUs from one host process (BASELINE config 2 behind the C ABI; SURVEY.md 8e: independent
// proofs, proof i -> device i mod n_dev, no data-path collective).  ctxs[d] / pks[d]: one context per device and
// the key set up (or loaded) on that context.  One host thread per device runs that device's share through the
// pipelined batch prover; results land in out_proofs in the caller's order.  z[i]: witness of proof i, a host
// pointer (z_on_device = 0; pinned memory uploads asynchronously) or a device pointer on the device of
// ctxs[i % n_dev] (z_on_device = 1).  Returns the first failing device's code (its message is in that ctx).
int32_t zkmi_groth16_prove_batch_multi(zkmi_ctx* const* ctxs, const zkmi_pk* const* pks, uint32_t n_dev, uint32_t n_proofs,
                                       const void* const* z, int32_t z_on_device, const uint8_t* r_bytes,
                                       const uint8_t* s_bytes, uint8_t* out_proofs) {
  if (!ctxs || !pks || n_dev == 0 || n_dev > 64 || (n_proofs && (!z || !r_bytes || !s_bytes || !out_proofs))) return ZKMI_ERR_BAD_ARG;
  for (uint32_t d = 0; d < n_dev; d++) {
    if (!ctxs[d] || !pks[d]) return ZKMI_ERR_BAD_ARG;
    for (uint32_t e = 0; e < d; e++)
      if (ctxs[e] == ctxs[d]) return ZKMI_ERR_BAD_ARG;  // a context serves one host thread at a time
    if (pks[d]->n_vars != pks[0]->n_vars || pks[d]->n_pub != pks[0]->n_pub || pks[d]->log_n != pks[0]->log_n)
      return ctxs[d]->fail(ZKMI_ERR_BAD_ARG, "the keys of a multi-device batch must be replicas of one key");
    // the two arrays are coupled by index only: a swapped or stale entry would run ctxs[d]'s streams over buffers,
    // tables and pinned flags of another context (another device)
    if (pks[d]->ctx != ctxs[d] || pks[d]->device != ctxs[d]->device)
      return ctxs[d]->fail(ZKMI_ERR_BAD_ARG, "pks[d] was not created on ctxs[d]");
  }
  // every share's inputs and outputs are laid out BEFORE any thread starts: the library is built without exceptions, so an
  // allocation that fails inside a worker would abort the host process from a C ABI call
  struct Share {
    std::vector<const void*> zz;
    std::vector<uint8_t> rr, ss, out;
  };
  std::vector<Share> sh(n_dev);
  for (uint32_t d = 0; d < n_dev; d++) {
    Share& q = sh[d];
    for (uint32_t i = d; i < n_proofs; i += n_dev) {
      q.zz.push_back(z[i]);
      q.rr.insert(q.rr.end(), r_bytes + 32ull * i, r_bytes + 32ull * i + 32);
      q.ss.insert(q.ss.end(), s_bytes + 32ull * i, s_bytes + 32ull * i + 32);
    }
    q.out.resize(192 * q.zz.size());
  }
  std::vector<int32_t> rc(n_dev, ZKMI_OK);
  auto share = [&](uint32_t d) {
    zkmi_ctx* ctx = ctxs[d];
    Share& q = sh[d];
    if (q.zz.empty()) return;
    if (hipSetDevice(ctx->device) != hipSuccess) {
      rc[d] = ctx->fail(ZKMI_ERR_HIP, "hipSetDevice");
      return;
    }
    rc[d] = prove_batch(ctx, pks[d], (uint32_t)q.zz.size(), q.zz.data(), z_on_device == 0, q.rr.data(), q.ss.data(), q.out.data());
    if (rc[d] != ZKMI_OK) return;
    for (size_t k = 0; k < q.zz.size(); k++) memcpy(out_proofs + 192ull * (d + k * n_dev), q.out.data() + 192 * k, 192);
  };
  if (n_dev == 1) {
    share(0);
    return rc[0];
  }
  std::vector<std::thread> th;
  th.reserve(n_dev);
  for (uint32_t d = 1; d < n_dev; d++) th.emplace_back(share, d);
  share(0);  // the calling thread drives device 0
  for (auto& t : th) t.join();
  for (uint32_t d = 0; d < n_dev; d++)
    if (rc[d] != ZKMI_OK) return rc[d];
  return ZKMI_OK;
}

int32_t zkmi_groth16_verify(const uint8_t* vk, uint32_t n_pub, const uint8_t* publics, const uint8_t proof[192]) {
  if (!vk || !proof || n_pub == 0 || (n_pub > 1 && !publics)) return ZKMI_ERR_BAD_ARG;
  G1Affine alpha, a, c;
  G2Affine beta, gamma, delta, b;
  if (!g1_from_wire(vk, &alpha, true) || !g2_from_wire(vk + 96, &beta, true) ||
      !g2_from_wire(vk + 288, &gamma, true) || !g2_from_wire(vk + 480, &delta, true))
    return ZKMI_ERR_NON_CANONICAL;
  // canonical encodings only (one encoding of infinity), then subgroup membership of every point
  // that enters a Miller loop: an on-curve point of small order would make the check malleable
  if (!g1_decompress(proof, &a) || !g2_decompress(proof + 48, &b) || !g1_decompress(proof + 144, &c))
    return ZKMI_ERR_NON_CANONICAL;
  if (!g1_in_subgroup(a) || !g2_in_subgroup(b) || !g1_in_subgroup(c)) return ZKMI_ERR_NON_CANONICAL;
  if (!g1_in_subgroup(alpha) || !g2_in_subgroup(beta) || !g2_in_subgroup(gamma) || !g2_in_subgroup(delta))
    return ZKMI_ERR_NON_CANONICAL;
  G1Affine ic0;
  if (!g1_from_wire(vk + 672, &ic0, true) || !g1_in_subgroup(ic0)) return ZKMI_ERR_NON_CANONICAL;
  G1XYZZ acc = G1XYZZ::from_affine(ic0);
  for (uint32_t j = 1; j < n_pub; j++) {
    G1Affine icj;
    if (!g1_from_wire(vk + 672 + 96ull * j, &icj, true) || !g1_in_subgroup(icj)) return ZKMI_ERR_NON_CANONICAL;
    if (!fr_is_canonical(publics + 32ull * (j - 1))) return ZKMI_ERR_NON_CANONICAL;
    uint32_t k[8];
    memcpy(k, publics + 32ull * (j - 1), 32);
    acc.add(scalar_mul(G1XYZZ::from_affine(icj), k, 8));
  }
  // e(-A,B) e(alpha,beta) e(acc,gamma) e(C,delta) == 1
  Fq12 f = miller_loop(a.neg(), b) * miller_loop(alpha, beta);
  f = f * miller_loop(acc.to_affine(), gamma) * miller_loop(c, delta);
  return final_exponentiation(f) == Fq12::one() ? ZKMI_OK : ZKMI_ERR_VERIFICATION;
}

}  // extern "C"

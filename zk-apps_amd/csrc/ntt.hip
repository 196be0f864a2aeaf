// zkmi — radix-2 NTT over the BLS12-381 scalar field on gfx950.
//
// Reference locus: none in /root/reference.  Semantics
// = ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place and the coset
// variants (generator 7), as restated in oracle/ntt.py.
//
// Element type: Fr28 = 10 signed 28-bit limbs (40 B), Montgomery R = 2^280,
// lazily reduced (field28.hpp): a butterfly is one 270-instruction product
// plus two carry-swept add/sub; values may grow by ~1.5 r per stage, which the
// 25 spare bits of the radix absorb, so nothing is reduced between stages.
//
// Kernel plan (LDS-staged butterflies).  A pass owns up to 10 consecutive
// butterfly stages: every workgroup stages a tile of 2^S x 2^Q elements in LDS
// (2^S strided sub-problem points x 2^Q adjacent columns), runs the S stages out
// of LDS with one barrier per stage, and writes the tile back.  N = 2^20 is two
// passes over HBM (1024 x 40 B contiguous tiles, then 1024 x 2 strided tiles of
// 80 KiB).  Two orderings avoid every bit-reversal copy inside the prover:
//   DIF (Gentleman-Sande): natural in  -> bit-reversed out   (inverse transforms)
//   DIT (Cooley-Tukey)   : bit-reversed in -> natural out    (forward transforms)
// and the coset / 1/N scalings are folded into the last pass of the DIF
// transforms (table indexed by position).  The natural-order public entry point
// adds one bit-reversal copy in front of a DIT transform.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "ntt.hpp"
#include "tune.hpp"

namespace zkmi {

namespace {

__device__ __forceinline__ void ld_words8(const uint32_t* p, uint32_t* w) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void st_words8(uint32_t* p, const uint32_t* w) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F>
__global__ void k_bitrev_copy(const F* __restrict__ in, F* __restrict__ out, int log_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  const uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0u;
  st28(out + r, ld28(in + i));
}

// One pass over stages [t0, t0+S).  DIF runs them high -> low, DIT low -> high.
// tw[k] = w^k (or w^-k), k < N/2.
// LOCAL_TW (transforms of one or two passes): the pass is a pure 2^S-point transform whose
// 2^(S-1) twiddles are staged in LDS once per workgroup; the cross terms w^(column * row) of
// the two-pass split are applied once per element ("twist") — at the load of the strided
// DIT pass, at the store of the strided DIF pass — instead of one global twiddle gather
// per butterfly.  Without LOCAL_TW (three passes, N > 2^20) twiddles are gathered per butterfly.
// post (optional): every output is multiplied by post[position].  canon_out (optional):
// outputs are written as canonical 32-byte integers instead of limbs (H MSM digits).
template <class F, bool DIF, bool LOCAL_TW, int THREADS>
__global__ void __launch_bounds__(THREADS)
k_ntt_pass(F* __restrict__ data, const F* __restrict__ tw, int log_n, int t0, int S, int Q,
           const F* __restrict__ post, uint32_t* __restrict__ canon_out) {
  // blockIdx.y = transform of a batch: vectors of 2^log_n elements laid out back to back (same for canon_out)
  data += (size_t)blockIdx.y << log_n;
  if (canon_out) canon_out += ((size_t)blockIdx.y << log_n) * 8;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  F* tile = reinterpret_cast<F*>(lds_raw);
  const uint32_t tile_n = 1u << (S + Q);
  F* ctw = tile + tile_n;  // LOCAL_TW: w_{2^S}^k, k < 2^(S-1)
  const uint32_t blk = blockIdx.x;
  // tile element L -> global index g:
  //   t0 == 0 : g = blk * tile_n + L
  //   t0 >  0 : L = e * 2^Q + c ; g = hi << (t0+S) | e << t0 | mid << Q | c, blk = hi * 2^(t0-Q) + mid
  const uint32_t mid_bits = (t0 > 0) ? (uint32_t)(t0 - Q) : 0u;
  const uint32_t mid = blk & ((1u << mid_bits) - 1u);
  const uint32_t hi = blk >> mid_bits;
  const uint32_t qeff = (t0 > 0) ? (uint32_t)Q : 0u;
  auto gindex = [&](uint32_t L) -> uint32_t {
    if (t0 == 0) return blk * tile_n + L;
    const uint32_t e = L >> Q, c = L & ((1u << Q) - 1u);
    return (hi << (t0 + S)) | (e << t0) | (mid << Q) | c;
  };
  // w^(column * rev_S(row)) for the element at tile position L of a strided pass
  auto twist = [&](uint32_t L) -> F {
    const uint32_t e = L >> Q, c = L & ((1u << Q) - 1u);
    const uint32_t col = (mid << Q) | c;
    // (root of order 2^(t0+S): the shift is zero for the last pass of a plan, non-zero for the middle pass of three)
    const uint32_t ex = (col * (__brev(e) >> (32 - S))) << (log_n - t0 - S);
    const uint32_t halfn = 1u << (log_n - 1);
    F f = ld28(tw + (ex & (halfn - 1u)));
    return (ex & halfn) ? f.neg() : f;
  };
  if (LOCAL_TW)
    for (uint32_t k = threadIdx.x; k < (1u << (S - 1)); k += THREADS) ctw[k] = ld28(tw + ((size_t)k << (log_n - S)));
  for (uint32_t L = threadIdx.x; L < tile_n; L += THREADS) {
    F v = ld28(data + gindex(L));
    if (LOCAL_TW && !DIF && t0 > 0) v = v * twist(L);
    tile[L] = v;
  }
  __syncthreads();

  const uint32_t half = tile_n >> 1;
  for (int s = 0; s < S; s++) {
    const int u = DIF ? (S - 1 - s) : s;
    const int t = t0 + u;  // global stage: butterfly distance 2^t
    const uint32_t dist_log = (uint32_t)u + qeff;
    const uint32_t dist = 1u << dist_log;
    for (uint32_t b = threadIdx.x; b < half; b += THREADS) {
      const uint32_t lo = b & (dist - 1u);
      const uint32_t L0 = ((b >> dist_log) << (dist_log + 1)) | lo;
      const uint32_t L1 = L0 | dist;
      F w;
      if (LOCAL_TW) {
        w = ctw[(lo >> qeff) << (S - 1 - u)];
      } else {
        const uint32_t j = gindex(L0) & ((1u << t) - 1u);
        w = ld28(tw + ((size_t)j << (log_n - 1 - t)));
      }
      const F x = tile[L0];
      if (DIF) {
        const F y = tile[L1];
        tile[L0] = x + y;
        tile[L1] = x.sub_lazy(y) * w;
      } else {
        const F y = tile[L1] * w;
        tile[L0] = x + y;
        tile[L1] = x - y;
      }
    }
    __syncthreads();
  }

  for (uint32_t L = threadIdx.x; L < tile_n; L += THREADS) {
    const uint32_t g = gindex(L);
    F v = tile[L];
    if (LOCAL_TW && DIF && t0 > 0) v = v * twist(L);
    if (post) v = v * ld28(post + g);
    if (canon_out) {
      uint32_t w[8];
      v.to_canonical(w);
      st_words8(canon_out + (size_t)g * 8, w);
    } else {
      st28(data + g, v);
    }
  }
}

#ifdef ZKMI_EXPERIMENTS
#include "ntt_exp.hpp"  // register-blocked / one-wave pass kernels: A/B library only
#endif  // ZKMI_EXPERIMENTS

template <class F>
__global__ void __launch_bounds__(256)
k_mul_table(F* __restrict__ a, const F* __restrict__ b, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st28(a + i, ld28(a + i) * ld28(b + i));
}

template <class F>
__global__ void __launch_bounds__(256)
k_scale(F* __restrict__ a, F s, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st28(a + i, ld28(a + i) * s);
}

// out[i] = first * base^i, i < n: thread t starts at base^(16 t) by square-and-multiply
template <class F>
__global__ void __launch_bounds__(256)
k_power_table(F* __restrict__ out, F base, F first, uint32_t n) {
  const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
  if (i0 >= n) return;
  F v = first, sq = base;
  for (uint32_t e = i0; e; e >>= 1) {
    if (e & 1u) v = v * sq;
    sq = sq * sq;
  }
  for (uint32_t k = 0; k < 16u && i0 + k < n; k++) {
    st28(out + i0 + k, v);
    v = v * base;
  }
}

// canonical 32-byte integers <-> limbs (Montgomery R = 2^280)
template <class F>
__global__ void __launch_bounds__(256)
k_from_canonical(const uint32_t* __restrict__ in, F* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  ld_words8(in + (size_t)i * 8, w);
  st28(out + i, F::from_canonical(w));
}
template <class F>
__global__ void __launch_bounds__(256)
k_to_canonical(const F* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  ld28(in + i).to_canonical(w);
  st_words8(out + (size_t)i * 8, w);
}

}  // namespace

// ---------------------------------------------------------------------------
// host-side driver
// ---------------------------------------------------------------------------
Fr fr_root_of_unity(int log_n) {
  // 7^((r-1)/2^32), then square down to order 2^log_n
  static const uint32_t ROOT_2_32[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                                        0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
  Fr w;
  for (int i = 0; i < 8; i++) w.l[i] = ROOT_2_32[i];
  w = w.to_mont();
  for (int i = 32; i > log_n; i--) w = w.sqr();
  return w;
}

NttField<Fr28>::Host NttField<Fr28>::root_max() { return fr_root_of_unity(32); }
NttField<BnFr28>::Host NttField<BnFr28>::root_max() {
  // halo2curves bn256::Fr::ROOT_OF_UNITY = 7^((r-1)/2^28)
  static const uint32_t ROOT_2_28[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du,
                                        0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
  BnFr w;
  for (int i = 0; i < 8; i++) w.l[i] = ROOT_2_28[i];
  return w.to_mont();
}

template <class H>
static H host_from_u64(uint64_t v) {
  H a = H::zero();
  a.l[0] = (uint32_t)v;
  a.l[1] = (uint32_t)(v >> 32);
  return a.to_mont();
}
template <class F, class H>
static F to28(const H& a) {
  H c = a.from_mont();
  return F::from_canonical(c.l);
}

template <class F>
NttDomainT<F>::~NttDomainT() {
  F* ptrs[] = {tw_fwd, tw_inv, coset_fwd, coset_inv_n, rev_coset_n, rev_coset_inv_n, n_inv, scratch};
  for (F* p : ptrs)
    if (p) (void)hipFree(p);
}

template <class F>
hipError_t NttDomainT<F>::init(int log_n_, hipStream_t stream) {
  using H = typename NttField<F>::Host;
  if (log_n_ > NttField<F>::TWO_ADICITY) return hipErrorInvalidValue;
  log_n = log_n_;
  const uint32_t n = 1u << log_n;
  const uint32_t half = n > 1 ? n / 2 : 1;
  hipError_t e;
  F** alloc_n[] = {&coset_fwd, &coset_inv_n, &rev_coset_n, &rev_coset_inv_n, &scratch};
  if ((e = hipMalloc(&tw_fwd, sizeof(F) * half)) != hipSuccess) return e;
  if ((e = hipMalloc(&tw_inv, sizeof(F) * half)) != hipSuccess) return e;
  if ((e = hipMalloc(&n_inv, sizeof(F))) != hipSuccess) return e;
  for (F** p : alloc_n)
    if ((e = hipMalloc(p, sizeof(F) * n)) != hipSuccess) return e;
  H w = NttField<F>::root_max();
  for (int i = NttField<F>::TWO_ADICITY; i > log_n; i--) w = w.sqr();
  const H g = host_from_u64<H>(7);
  const H ninv = host_from_u64<H>(n).inv();
  const F one28 = F::one(), ninv28 = to28<F, H>(ninv);
  n_inv_host = ninv28;
  const int T = 256;
  auto blocks = [&](uint32_t cnt) { return (cnt + 16 * T - 1) / (16 * T); };
  hipLaunchKernelGGL(k_power_table<F>, dim3(blocks(half)), dim3(T), 0, stream, tw_fwd, to28<F, H>(w), one28, half);
  hipLaunchKernelGGL(k_power_table<F>, dim3(blocks(half)), dim3(T), 0, stream, tw_inv, to28<F, H>(w.inv()), one28, half);
  hipLaunchKernelGGL(k_power_table<F>, dim3(blocks(n)), dim3(T), 0, stream, coset_fwd, to28<F, H>(g), one28, n);
  hipLaunchKernelGGL(k_power_table<F>, dim3(blocks(n)), dim3(T), 0, stream, coset_inv_n, to28<F, H>(g.inv()), ninv28, n);
  // position-indexed tables for bit-reversed coefficient order: entry p <- index rev(p)
  hipLaunchKernelGGL(k_power_table<F>, dim3(blocks(n)), dim3(T), 0, stream, scratch, to28<F, H>(g), ninv28, n);
  hipLaunchKernelGGL(k_bitrev_copy<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, scratch, rev_coset_n, log_n);
  hipLaunchKernelGGL(k_bitrev_copy<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, coset_inv_n, rev_coset_inv_n, log_n);
  if ((e = hipMemcpyAsync(n_inv, &ninv28, sizeof(F), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  return hipGetLastError();
}

// stages [0, log_n) split into passes of <= 10 stages; pass k of a DIT transform
// covers the low stages first, of a DIF transform the high stages first
// The product's pass kernel is k_ntt_pass (one butterfly per thread and barrier); a short pass (S < 10: the third pass of
// N >= 2^21) takes 2^(11-S) adjacent columns per tile when that still leaves >= 512 tiles: 2048-element tiles with 1024
// busy threads instead of 8-element tiles in 64-thread workgroups (N = 2^22: 1.62 -> 1.1 ms per transform).  That is mode 3
// of the A/B library's ZKMI_NTT_RB switch:
//   0 = the round-2 form: short strided passes in 2-column tiles
//   1 / 2 = register-blocked passes k_ntt_pass_rb with 8 / 4 elements per thread (256- / 512-thread workgroups),
//       2048-element tiles wherever the transform has them.  Measured in round 3 (profiles/r03/ntt_variants.txt): equal
//       to the plain kernel at N = 2^20 alone (0.25 ms per transform), slower below 2^18 (fewer, larger workgroups),
//       and the 8-element form costs the prover 2-3 % (one 256-VGPR wave per SIMD plus 105 KB of LDS leaves room for
//       one accumulation wave instead of three beside it)
//   4 / 5 = one-wave workgroups on 1024- / 512-element tiles (5: three passes of <= 7 stages at N = 2^20)
//   6 (round 6, the product's): mode 3 with HALF tiles where a pass has a full 2^11-element one -- 2^10 elements on 512 threads,
//       60 KiB of LDS, two workgroups per CU: 2^16 0.075 -> 0.063 ms, 2^20 0.292 -> 0.271, 2^22 1.205 -> 1.075 per transform
//       through the C ABI (profiles/r06/experiments/ntt_half_tiles_ab.txt)
static int ntt_rb_mode() { return ZK_TUNE("ZKMI_NTT_RB", 6); }

#ifdef ZKMI_EXPERIMENTS
template <class F, bool DIF, bool LTW, int LOGE, int LOGR = 0, bool ONEW = false>
static void launch_rb(dim3 grid, uint32_t tile_n, int S, hipStream_t stream, F* buf, const F* tw, int log_n, int t0, int Q,
                      const F* post, uint32_t* canon_out) {
  const size_t words = (size_t)F::NL * (tile_n + (tile_n >> 5)) +
                       (LTW ? (size_t)F::NL * ((1u << (S - 1)) + ((1u << (S - 1)) >> 5) + 1u) : 0u);
  hipLaunchKernelGGL((k_ntt_pass_rb<F, DIF, LTW, LOGE, LOGR, ONEW>), grid, dim3(tile_n >> (LOGE + LOGR)), words * sizeof(uint32_t), stream,
                     buf, tw, log_n, t0, S, Q, post, canon_out);
}
#endif

template <class F, bool DIF>
static hipError_t run_passes(F* buf, const F* tw, int log_n, const F* post, uint32_t* canon_out,
                             hipStream_t stream, uint32_t batch = 1) {
  const uint32_t n = 1u << log_n;
  int mode = ntt_rb_mode();
  if (mode == 5 && log_n < 9) mode = 3;  // mode 5 needs 512-element tiles
  const int rb = (mode == 1 || mode == 2 || mode == 4 || mode == 5) ? mode : 0;
  const int tile_log = mode == 5 ? 9 : mode == 4 ? 10 : 11;  // modes 4 / 5: 1024- / 512-element tiles in one-wave workgroups
  struct Pass {
    int t0, S, Q;
  } passes[4];
  int np = 0;
  // mode 5: passes of <= 9 stages, split evenly (N = 2^20: 7 + 7 + 6), every pass with the local-twiddle + twist scheme
  const int max_s = mode == 5 ? 9 : 10;
  const int want_np = (log_n + max_s - 1) / max_s;
  for (int t0 = 0; t0 < log_n;) {
    int S = log_n - t0;
    if (S > max_s) S = max_s;
    if (mode == 5) {
      const int left = want_np - np;  // passes still to come, this one included
      S = (log_n - t0 + left - 1) / left;
    }
    int Q;
    if (rb) {
      // tiles of 2^11 elements wherever the transform has them: a strided pass takes 2^Q adjacent columns (2^Q x 40 B
      // contiguous per access), the contiguous pass 2^Q whole sub-transforms
      Q = tile_log - S;
      const int room = (t0 > 0) ? t0 : (log_n - S);
      if (Q > room) Q = room;
    } else {
      // strided passes: 2 adjacent columns; the contiguous pass: 2 sub-transforms per
      // workgroup so that all 1024 threads own a butterfly in every stage
      Q = (t0 > 0) ? 1 : ((log_n > S) ? 1 : 0);
      // wide tiles for a short strided pass of a big transform (mode 3): 2^11 elements per tile, >= 512 tiles per vector
      if ((mode == 3 || mode == 6) && t0 > 0 && S < 10 && log_n - 11 >= 9) Q = (11 - S < t0) ? 11 - S : t0;
      // mode 6 (round 6): HALF tiles -- 2^10 elements (40 KiB + 20 KiB of twiddles) on 512 threads, so that TWO workgroups
      // share a CU's 160 KiB of LDS and one computes while the other waits at a barrier (the 2^11-element tile is the only
      // workgroup on its CU: each of its 10 barriers idles the CU)
      if (mode == 6 && S + Q == 11) Q -= 1;
    }
    passes[np++] = {t0, S, Q};
    t0 += S;
  }
  // Plans of three passes (N > 2^20) gather a twiddle per butterfly (it comes out of L2); the local-twiddle + twist scheme
  // also covers them since round 3 (the twist of a middle pass is a power of the root of order 2^(t0+S)) but measured 5 %
  // slower there (2^21: 0.549 vs 0.518 ms): ZKMI_NTT_LOCAL3=1 selects it; mode 5 always uses it.
  const bool local3 = ZK_TUNE("ZKMI_NTT_LOCAL3", 0) == 1;
  const bool local_tw = np <= 2 || mode == 5 || local3;
  for (int k = 0; k < np; k++) {
    const Pass& p = DIF ? passes[np - 1 - k] : passes[k];
    const bool last = (k == np - 1);
    const uint32_t tile_n = 1u << (p.S + p.Q);
    const uint32_t nblk = n / tile_n;
    const F* pp = last ? post : nullptr;
    uint32_t* co = last ? canon_out : nullptr;
    const dim3 grid(nblk, batch);
#ifdef ZKMI_EXPERIMENTS
    if (rb == 5) {  // tile_n == 512 by construction (log_n >= 9)
      launch_rb<F, DIF, true, 2, 1, true>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);  // 4 elements x 2 units per thread
      continue;
    }
    if (rb == 4 && tile_n == 1024) {
      if (local_tw) launch_rb<F, DIF, true, 2, 2>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
      else launch_rb<F, DIF, false, 2, 2>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
      continue;
    }
    if (rb && rb != 4 && rb != 5 && tile_n >= 512) {
      if (rb == 2) {
        if (local_tw) launch_rb<F, DIF, true, 2>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
        else launch_rb<F, DIF, false, 2>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
      } else {
        if (local_tw) launch_rb<F, DIF, true, 3>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
        else launch_rb<F, DIF, false, 3>(grid, tile_n, p.S, stream, buf, tw, log_n, p.t0, p.Q, pp, co);
      }
      continue;
    }
#endif
    const size_t lds = ((size_t)tile_n + (local_tw ? (1u << (p.S - 1)) : 0u)) * sizeof(F);
    if (tile_n == 1024 && mode == 6) {
      if (local_tw)
        hipLaunchKernelGGL((k_ntt_pass<F, DIF, true, 512>), grid, dim3(512), lds, stream, buf, tw, log_n, p.t0, p.S, p.Q, pp, co);
      else
        hipLaunchKernelGGL((k_ntt_pass<F, DIF, false, 512>), grid, dim3(512), lds, stream, buf, tw, log_n, p.t0, p.S, p.Q, pp, co);
    } else if (tile_n >= 1024) {
      if (local_tw)
        hipLaunchKernelGGL((k_ntt_pass<F, DIF, true, 1024>), grid, dim3(1024), lds, stream, buf, tw, log_n, p.t0, p.S,
                           p.Q, pp, co);
      else
        hipLaunchKernelGGL((k_ntt_pass<F, DIF, false, 1024>), grid, dim3(1024), lds, stream, buf, tw, log_n, p.t0,
                           p.S, p.Q, pp, co);
    } else if (local_tw) {
      hipLaunchKernelGGL((k_ntt_pass<F, DIF, true, 64>), grid, dim3(64), lds, stream, buf, tw, log_n, p.t0, p.S, p.Q,
                         pp, co);
    } else {
      hipLaunchKernelGGL((k_ntt_pass<F, DIF, false, 64>), grid, dim3(64), lds, stream, buf, tw, log_n, p.t0, p.S, p.Q,
                         pp, co);
    }
  }
  return hipGetLastError();
}

template <class F>
hipError_t NttDomainT<F>::inverse_to_rev(F* d, const F* post_table, uint32_t* canon_out, hipStream_t st, uint32_t batch) {
  if (log_n == 0) {
    if (batch != 1) return hipErrorInvalidValue;
    // single element: only the post factor / output format applies
    if (post_table) hipLaunchKernelGGL(k_mul_table<F>, dim3(1), dim3(256), 0, st, d, post_table, 1u);
    if (canon_out) hipLaunchKernelGGL(k_to_canonical<F>, dim3(1), dim3(256), 0, st, d, canon_out, 1u);
    return hipGetLastError();
  }
  return run_passes<F, true>(d, tw_inv, log_n, post_table, canon_out, st, batch);
}

template <class F>
hipError_t NttDomainT<F>::forward_from_rev(F* d, hipStream_t st, uint32_t batch) {
  if (log_n == 0) return hipSuccess;
  return run_passes<F, false>(d, tw_fwd, log_n, nullptr, nullptr, st, batch);
}

// natural order in and out (public entry point)
template <class F>
hipError_t NttDomainT<F>::transform(F* d_data, bool inverse, bool coset, hipStream_t stream) {
  const uint32_t n = 1u << log_n;
  const int T = 256;
  if (coset && !inverse)
    hipLaunchKernelGGL(k_mul_table<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, coset_fwd, n);
  if (log_n > 0) {
    hipLaunchKernelGGL(k_bitrev_copy<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, scratch, log_n);
    hipError_t e = run_passes<F, false>(scratch, inverse ? tw_inv : tw_fwd, log_n, nullptr, nullptr, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_data, scratch, sizeof(F) * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
  }
  if (inverse) {
    if (coset) {
      hipLaunchKernelGGL(k_mul_table<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, coset_inv_n, n);
    } else {
      hipLaunchKernelGGL(k_scale<F>, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, n_inv_host, n);
    }
  }
  return hipGetLastError();
}

template <class F>
hipError_t ntt_from_canonical(const uint32_t* d_in, F* d_out, uint32_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_from_canonical<F>, dim3((n + 63) / 64), dim3(64), 0, s, d_in, d_out, n);  // one-wave workgroups
  return hipGetLastError();
}
template <class F>
hipError_t ntt_to_canonical(const F* d_in, uint32_t* d_out, uint32_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_to_canonical<F>, dim3((n + 255) / 256), dim3(256), 0, s, d_in, d_out, n);
  return hipGetLastError();
}
template <class F>
hipError_t ntt_mul_table(F* d, const F* table, uint32_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_mul_table<F>, dim3((n + 255) / 256), dim3(256), 0, s, d, table, n);
  return hipGetLastError();
}

#ifdef ZKMI_EXPERIMENTS
template <class F>
static hipError_t rb_enable_big_lds() {
  const void* fns[] = {reinterpret_cast<const void*>(k_ntt_pass_rb<F, true, true, 3>),  reinterpret_cast<const void*>(k_ntt_pass_rb<F, false, true, 3>),
                       reinterpret_cast<const void*>(k_ntt_pass_rb<F, true, false, 3>), reinterpret_cast<const void*>(k_ntt_pass_rb<F, false, false, 3>),
                       reinterpret_cast<const void*>(k_ntt_pass_rb<F, true, true, 2>),  reinterpret_cast<const void*>(k_ntt_pass_rb<F, false, true, 2>),
                       reinterpret_cast<const void*>(k_ntt_pass_rb<F, true, false, 2>), reinterpret_cast<const void*>(k_ntt_pass_rb<F, false, false, 2>),
                       reinterpret_cast<const void*>((k_ntt_pass_rb<F, true, true, 2, 2>)),  reinterpret_cast<const void*>((k_ntt_pass_rb<F, false, true, 2, 2>)),
                       reinterpret_cast<const void*>((k_ntt_pass_rb<F, true, false, 2, 2>)), reinterpret_cast<const void*>((k_ntt_pass_rb<F, false, false, 2, 2>)),
                       reinterpret_cast<const void*>((k_ntt_pass_rb<F, true, true, 2, 1, true>)), reinterpret_cast<const void*>((k_ntt_pass_rb<F, false, true, 2, 1, true>))};
  for (const void* f : fns) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
#endif

hipError_t ntt_enable_big_lds() {
  const void* fns[] = {reinterpret_cast<const void*>(k_ntt_pass<Fr28, true, true, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, false, true, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, true, false, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, false, false, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, true, true, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, false, true, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, true, false, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, false, false, 1024>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, true, true, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, false, true, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, true, false, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<Fr28, false, false, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, true, true, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, false, true, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, true, false, 512>),
                       reinterpret_cast<const void*>(k_ntt_pass<BnFr28, false, false, 512>)};
  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
#ifdef ZKMI_EXPERIMENTS
  hipError_t e = rb_enable_big_lds<Fr28>();
  if (e == hipSuccess) e = rb_enable_big_lds<BnFr28>();
  return e;
#else
  return hipSuccess;
#endif
}

template struct NttDomainT<Fr28>;
template struct NttDomainT<BnFr28>;
template hipError_t ntt_from_canonical<Fr28>(const uint32_t*, Fr28*, uint32_t, hipStream_t);
template hipError_t ntt_to_canonical<Fr28>(const Fr28*, uint32_t*, uint32_t, hipStream_t);
template hipError_t ntt_mul_table<Fr28>(Fr28*, const Fr28*, uint32_t, hipStream_t);
template hipError_t ntt_from_canonical<BnFr28>(const uint32_t*, BnFr28*, uint32_t, hipStream_t);
template hipError_t ntt_to_canonical<BnFr28>(const BnFr28*, uint32_t*, uint32_t, hipStream_t);
template hipError_t ntt_mul_table<BnFr28>(BnFr28*, const BnFr28*, uint32_t, hipStream_t);

}  // namespace zkmi

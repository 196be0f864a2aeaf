// zkmi — Pippenger bucket-method MSM on gfx950, templated on the base field
// (F = Fq -> G1, F = Fq2 -> G2).
//
// Reference locus: none in /root/reference.  The
// value computed is sum_i s_i * P_i, the same group element
// ark_ec::VariableBaseMSM::msm_bigint / halo2curves::msm::best_multiexp
// return; any correct bucket schedule is bit-exact after affine normalisation.
//
// HBM layout
//   scalars   : n x 32 B canonical little-endian integers < r (NOT Montgomery)
//   bases     : n x Affine<F>, F = 14 x 28-bit limbs (field28.hpp), (0,0) = infinity
//   blockhist : nwin*nch*nb x u32  per-(window, chunk) tile histograms -> tile base slots
//   count/begin : nwin*nb x u32    points per bucket / first slot in sorted[]
//   sorted    : nwin*n x u32       point index | sign<<31, grouped by bucket
//   buckets   : nwin*nb x XYZZ<F>
//   segsum / segw : nwin*nb/SEG x XYZZ<F>
//   partial   : nwin*(1+log2(nb/SEG)) x XYZZ<host field>   -> host
//
// Kernel chain (all on one stream):
//   1 k_bucket_pass<false>  (window, chunk) tile: carry-free signed digits, LDS histogram
//                           (2^15 u32 counters = 128 KiB of the CU's 160 KiB LDS)
//   2 k_bucket_totals / k_window_scan / k_bucket_bases   prefix sums -> slots
//   3 k_bucket_pass<true>   same tiles: LDS cursors (ds_add_rtn) -> sorted[]   (bucket scatter)
//   4 k_accum_g1_nc / k_accum_g2_nc   thread (lane pair) per bucket : gather affine points, XYZZ mixed adds   <- dominant
//     k_accum_heavy[_g2_split]  workgroups per heavy bucket (the last one of a split bucket adds the sub-range sums),
//     k_accum_redo[_g2_split]   buckets that met P + P, with the complete addition
//   5 k_segreduce thread/2..16 buckets : running-sum  sum (i+1) B_i  and  sum B_i
//   6 k_treesum   block/(window, job[, slice]) : plain sums (LDS tree) of segw, and of
//                 segsum over {t : bit j of t set}; small plans: slices + k_treesum_final
//   host: window_w = P[w][0] + SEG * sum_j 2^j P[w][1+j];  result = sum_w 2^(c w) window_w
#pragma once
#include <stdlib.h>
#include <functional>
#include <type_traits>
#include <vector>
#include "curve.hpp"
#include "host_pool.hpp"
#include "msm.hpp"

namespace zkmi {

// segment arrays: 16-bucket segments for big plans, down to 1-bucket segments for plans of <= 2^16 buckets
static inline uint64_t msm_max_segments(uint64_t buckets) { return (buckets / 16 > (1u << 16) ? buckets / 16 : (1u << 16)) + 1; }
constexpr int MSM_TREE_T = 128;  // k_treesum block: 128 x XYZZ<Fq2> = 56 KiB LDS
constexpr uint32_t MSM_STAGE_PTS = 16384;  // slices of one slot's tree-sum job lists (k_treesum_q; the one-lane k_treesum uses the first MSM_STAGE_PTS_1)
constexpr uint32_t MSM_STAGE_PTS_1 = 4096;
constexpr uint32_t MSM_HEAVY = 256;  // load-ordering key range; the heavy threshold itself is plan.heavy_thr

template <class T>
__device__ __forceinline__ T load_vec(const T* p) {
  static_assert(sizeof(T) % 16 == 0, "16-byte multiple");
  T r;
  const uint4* s = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
  return r;
}
template <class T>
__device__ __forceinline__ void store_vec(T* p, const T& v) {
  static_assert(sizeof(T) % 16 == 0, "16-byte multiple");
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;


// G2 accumulation with every Fq2 value split across a lane pair (field28.hpp Fq2P):
// two adjacent lanes own one bucket; per-lane state is that of a G1 addition, so the
// kernel runs at 2 waves per SIMD instead of 1.  Memory layouts are the unsplit ones
// (x.c0, x.c1, y.c0, y.c1, ...): a lane simply reads / writes its own components.
__device__ __forceinline__ Fq28 ld_comp(const Fq28* p) {
  Fq28 r;
  const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const uint2 w = q[i];
    r.l[2 * i] = (int32_t)w.x;
    r.l[2 * i + 1] = (int32_t)w.y;
  }
  return r;
}
__device__ __forceinline__ void st_comp(Fq28* p, const Fq28& v) {
  uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
  for (int i = 0; i < 7; i++) q[i] = make_uint2((uint32_t)v.l[2 * i], (uint32_t)v.l[2 * i + 1]);
}


// ---------------------------------------------------------------------------------------------
// Call-free accumulation kernels.  XYZZ::madd keeps its rare doubling case in an out-of-line function;
// a call inside the hot kernel costs more than code size: the kernel's register count becomes the
// callee's (231-248 VGPRs whatever __launch_bounds__ asks for) and the frame needs scratch.  Here the
// doubling case (two equal points in one bucket: repeated bases with equal digits) never happens in
// the kernel: the lane appends its bucket to a redo list and stops; k_accum_redo recomputes the
// listed buckets with the complete addition afterwards.  The list holds bucket ids, so it cannot
// overflow (capacity = number of buckets).
// ---------------------------------------------------------------------------------------------
// acc += p for acc != O, p != O; returns false when the sum needs the complete group law (p = +-acc):
// the caller hands the bucket to k_accum_redo.  No state besides acc: the loop around it has one path.
// negmask = 0xffffffff adds -p instead of p (the sign of the digit): folded into the lazy difference R = +-(y zzz) - Y
template <class F>
__device__ __forceinline__ bool madd_generic(XYZZ<F>& acc, const Affine<F>& p, uint32_t negmask = 0) {
  F pp_ = f_sub_lazy(p.x * acc.zz, acc.x);
  F r = f_signed_sub_lazy(p.y * acc.zzz, negmask, acc.y);
  F pp = pp_.sqr();
  F rr = r.sqr();
  if (pp.is_zero()) return false;
  F ppp = pp_ * pp;
  acc.zz = acc.zz * pp;
  acc.zzz = acc.zzz * ppp;
  F q = acc.x * pp;
  acc.x = f_x3(rr, ppp, q);
  acc.y = f_mul_sub_mul(r, f_sub_lazy(q, acc.x), acc.y, ppp);
  return true;
}
// device table entries at infinity are exact zeros (k_bases_convert, k_build_table); one limb decides for all but
// 2^-28 of the finite entries, the full test runs behind that branch
template <class P>
__device__ __forceinline__ bool affine_is_zero_words(const Affine<Fp28<P>>& p) {
  if (p.y.l[0] != 0) return false;
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < P::NL; i++) o |= (uint32_t)p.x.l[i] | (uint32_t)p.y.l[i];
  return o == 0;
}

// G1 (and BN254 G1): thread per bucket, next table entry prefetched through ONE LDS buffer per wave
// (the entry is read into registers at the top of the iteration, so the buffer is free for the next
// direct-to-LDS load straight away): 7 KB per wave.
// What one launch accumulates: one MSM (table, bucket array, redo list + the digit sort it reads), or up to four MSMs of
// the same shape, one per blockIdx.y -- the A, B1, L queries of a proof over the sort of z and the H query over the sort of
// h.  One small proof uses the second form: 256-wave grids on separate streams did not start together (a kernel that
// cannot place all its workgroups at once holds its dispatch pipe, and the other streams of that pipe wait).
constexpr int MSM_MULTI_MAX = 4;
struct SortView {
  const uint32_t *begin, *count, *perm, *sorted;
  uint32_t heavy_thr;
};
template <class F, bool MULTI>
struct AccumArgs;
template <class F>
struct AccumArgs<F, false> {
  const Affine<F>* bases_;
  XYZZ<F>* buckets_;
  uint32_t* redo_;
  SortView sort_;
  __device__ __forceinline__ const Affine<F>* bases() const { return bases_; }
  __device__ __forceinline__ XYZZ<F>* buckets() const { return buckets_; }
  __device__ __forceinline__ uint32_t* redo() const { return redo_; }
  __device__ __forceinline__ const SortView& sort() const { return sort_; }
};
template <class F>
struct AccumArgs<F, true> {
  const Affine<F>* bases_[MSM_MULTI_MAX];
  XYZZ<F>* buckets_[MSM_MULTI_MAX];
  uint32_t* redo_[MSM_MULTI_MAX];
  SortView sort_[MSM_MULTI_MAX];
  __device__ __forceinline__ const Affine<F>* bases() const { return bases_[blockIdx.y]; }
  __device__ __forceinline__ XYZZ<F>* buckets() const { return buckets_[blockIdx.y]; }
  __device__ __forceinline__ uint32_t* redo() const { return redo_[blockIdx.y]; }
  __device__ __forceinline__ const SortView& sort() const { return sort_[blockIdx.y]; }
};
// INTO: the bucket array already holds the sums of an earlier MSM over the same bucket set (another query whose result is
// only ever ADDED to this one's: the prover's L and H queries both end up in C).  The accumulator then
// starts from the bucket's value instead of the first entry, an empty entry list leaves the bucket alone, and one
// reduction serves both MSMs.  A bucket the earlier MSM left EMPTY (exact zeros; probability e^-(mean load)) needs no
// test of its own: with acc = (0, 0, 0, 0) the first addition computes P = x zz - X = 0, which is the "needs the complete
// group law" exit below -- the bucket goes to the redo pass, whose complete addition starts from infinity.
// (Any early exit through the redo list IN FRONT of the loop made the register allocator spill 120 dwords inside the
// loop; this form spills 24 -- the plain kernel's 0 is a lucky draw at exactly the 168 registers of three waves per SIMD.)
template <class F, int W, int BW, bool MULTI = false, bool INTO = false>
__global__ void __launch_bounds__(64 * BW, W)
k_accum_g1_nc(const AccumArgs<F, MULTI> args, uint32_t total_buckets) {
  const Affine<F>* __restrict__ const bases = args.bases();
  XYZZ<F>* __restrict__ const buckets = args.buckets();
  uint32_t* __restrict__ const redo = args.redo();
  const uint32_t* __restrict__ const begin = args.sort().begin;
  const uint32_t* __restrict__ const count = args.sort().count;
  const uint32_t* __restrict__ const perm = args.sort().perm;
  const uint32_t* __restrict__ const sorted = args.sort().sorted;
  const uint32_t heavy_thr = args.sort().heavy_thr;
  constexpr int CHUNKS = sizeof(Affine<F>) / 16;  // 7 (BLS12-381 Fq), 5 (BN254 Fq)
  __shared__ uint4 tile[BW][CHUNKS][64];          // [wave][chunk][lane]
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (t >= total_buckets) return;
  const uint32_t b = perm[t];  // lanes of a wave own buckets of near-equal load
  const uint32_t cnt = count[b];
  if (cnt > heavy_thr) return;  // k_accum_heavy owns it
  const uint32_t beg = begin[b], end = beg + cnt;
  auto fetch = [&](uint32_t v) {
    const char* src = reinterpret_cast<const char*>(bases + (v & 0x7fffffffu));
#pragma unroll
    for (int q = 0; q < CHUNKS; q++)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 16 * q), (lds_ptr_t)&tile[wave][q][0], 16, 0, 0);
  };
  auto take = [&](Affine<F>& p) {  // LDS -> registers; the compiler waits for the outstanding LDS-DMA first
    uint4* d = reinterpret_cast<uint4*>(&p);
#pragma unroll
    for (int q = 0; q < CHUNKS; q++) d[q] = tile[wave][q][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have left the LDS: the buffer may be refilled
  };
  XYZZ<F> acc;
  uint32_t j = beg;
  // first entry that is not the point at infinity starts the accumulator (plain loads: runs once per bucket)
  for (;; j++) {
    if (j >= end) {
      if constexpr (!INTO) store_vec(buckets + b, XYZZ<F>::infinity());  // (INTO: nothing to add, the bucket keeps its sum)
      return;
    }
    const uint32_t v = sorted[j];
    Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
    if (affine_is_zero_words(p)) continue;
    if constexpr (INTO) {
      acc = load_vec(buckets + b);  // entry j itself is added by the loop below
      break;
    }
    if (v >> 31) p.y = p.y.neg();
    acc.x = p.x;
    acc.y = p.y;
    acc.zz = F::one();
    acc.zzz = F::one();
    j++;
    break;
  }
  uint32_t v_cur = 0, v_next = 0;
  if (j < end) {
    v_cur = sorted[j];
    fetch(v_cur);
    if (j + 1 < end) v_next = sorted[j + 1];
  }
  for (; j < end; j++) {
    Affine<F> p;
    take(p);
    const uint32_t v = v_cur;
    if (j + 1 < end) {
      fetch(v_next);
      v_cur = v_next;
      if (j + 2 < end) v_next = sorted[j + 2];
    }
    if (affine_is_zero_words(p)) continue;
    if (!madd_generic(acc, p, 0u - (v >> 31))) {
      // doubling or cancellation: k_accum_redo recomputes the bucket (INTO: from the value it still holds -- nothing
      // has been written)
      redo[1 + atomicAdd(redo, 1u)] = b;
      return;
    }
  }
  store_vec(buckets + b, acc);
}
#ifdef ZKMI_EXPERIMENTS
}  // namespace zkmi
#include "msm_impl_exp.hpp"  // retired kernel generations: A/B library only
namespace zkmi {
#endif
// acc += o for acc, o != O; false when the sum needs the complete group law (o = +-acc): same contract as madd_generic
template <class F>
__device__ __forceinline__ bool add_generic(XYZZ<F>& a, const XYZZ<F>& o) {
  F u1 = a.x * o.zz;
  F u2 = o.x * a.zz;
  F s1 = a.y * o.zzz;
  F s2 = o.y * a.zzz;
  F pp_ = f_sub_lazy(u2, u1);
  F r = f_sub_lazy(s2, s1);
  F pp = pp_.sqr();
  F rr = r.sqr();
  if (pp.is_zero()) return false;
  F ppp = pp_ * pp;
  F q = u1 * pp;
  F x3 = f_x3(rr, ppp, q);
  a.y = f_mul_sub_mul(r, f_sub_lazy(q, x3), s1, ppp);
  a.x = x3;
  a.zz = a.zz * o.zz * pp;
  a.zzz = a.zzz * o.zzz * ppp;
  return true;
}
// Two adjacent lanes per bucket (even lane: entries beg, beg + 2, ...; odd lane: beg + 1, beg + 3, ...), the even lane adds
// the two partial sums at the end.  For ONE small proof: its accumulation launch is 0.3 ms of chip time but lasts as long as
// the fullest bucket's chain of dependent additions (25-35 of them, 0.65 ms); two lanes halve the chain (section 4.11).
// Always the multi form (table / bucket array / redo list / sort per blockIdx.y); a doubling in either half or in the
// final addition sends the bucket to the redo list like the one-lane kernel does.
template <class F, int BW>
__global__ void __launch_bounds__(64 * BW, 2)
k_accum_g1_split2(const AccumArgs<F, true> args, uint32_t total_buckets) {
  const Affine<F>* __restrict__ const bases = args.bases();
  XYZZ<F>* __restrict__ const buckets = args.buckets();
  uint32_t* __restrict__ const redo = args.redo();
  const uint32_t* __restrict__ const begin = args.sort().begin;
  const uint32_t* __restrict__ const count = args.sort().count;
  const uint32_t* __restrict__ const perm = args.sort().perm;
  const uint32_t* __restrict__ const sorted = args.sort().sorted;
  const uint32_t heavy_thr = args.sort().heavy_thr;
  constexpr int CHUNKS = sizeof(Affine<F>) / 16;
  __shared__ uint4 tile[BW][CHUNKS][64];  // [wave][chunk][lane]
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = gt >> 1, sub = gt & 1u;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (t >= total_buckets) return;  // pair-uniform from here on
  const uint32_t b = perm[t];
  const uint32_t cnt = count[b];
  if (cnt > heavy_thr) return;  // k_accum_heavy owns it
  const uint32_t beg = begin[b], end = beg + cnt;
  auto fetch = [&](uint32_t v) {
    const char* src = reinterpret_cast<const char*>(bases + (v & 0x7fffffffu));
#pragma unroll
    for (int q = 0; q < CHUNKS; q++)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 16 * q), (lds_ptr_t)&tile[wave][q][0], 16, 0, 0);
  };
  auto take = [&](Affine<F>& p) {
    uint4* d = reinterpret_cast<uint4*>(&p);
#pragma unroll
    for (int q = 0; q < CHUNKS; q++) d[q] = tile[wave][q][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  XYZZ<F> acc = XYZZ<F>::infinity();
  bool has = false, bad = false;
  uint32_t j = beg + sub;
  for (; j < end; j += 2) {  // first entry of this lane that is not the point at infinity starts its accumulator
    const uint32_t v = sorted[j];
    Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
    if (affine_is_zero_words(p)) continue;
    if (v >> 31) p.y = p.y.neg();
    acc.x = p.x;
    acc.y = p.y;
    acc.zz = F::one();
    acc.zzz = F::one();
    has = true;
    j += 2;
    break;
  }
  uint32_t v_cur = 0, v_next = 0;
  if (j < end) {
    v_cur = sorted[j];
    fetch(v_cur);
    if (j + 2 < end) v_next = sorted[j + 2];
  }
  for (; j < end; j += 2) {
    Affine<F> p;
    take(p);
    const uint32_t v = v_cur;
    if (j + 2 < end) {
      fetch(v_next);
      v_cur = v_next;
      if (j + 4 < end) v_next = sorted[j + 4];
    }
    if (affine_is_zero_words(p)) continue;
    if (!madd_generic(acc, p, 0u - (v >> 31))) {
      bad = true;
      break;
    }
  }
  // the partner's state and partial sum
  const uint32_t flags = (has ? 1u : 0u) | (bad ? 2u : 0u);
  const uint32_t pflags = (uint32_t)__shfl_xor((int)flags, 1);
  XYZZ<F> o;
#pragma unroll
  for (int i = 0; i < F::NL; i++) {
    o.x.l[i] = __shfl_xor(acc.x.l[i], 1);
    o.y.l[i] = __shfl_xor(acc.y.l[i], 1);
    o.zz.l[i] = __shfl_xor(acc.zz.l[i], 1);
    o.zzz.l[i] = __shfl_xor(acc.zzz.l[i], 1);
  }
  if (sub != 0) return;
  bool redo_it = ((flags | pflags) & 2u) != 0;
  if (!redo_it) {
    if (has && (pflags & 1u)) redo_it = !add_generic(acc, o);
    else if (!has) acc = (pflags & 1u) ? o : XYZZ<F>::infinity();
  }
  if (redo_it)
    redo[1 + atomicAdd(redo, 1u)] = b;  // k_accum_redo recomputes the bucket with the complete addition
  else
    store_vec(buckets + b, acc);
}

// G2, lane pair per bucket (see k_accum_g2_split)
template <int W, int BW>
__global__ void __launch_bounds__(64 * BW, W)
k_accum_g2_nc(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ begin,
              const uint32_t* __restrict__ count, const uint32_t* __restrict__ perm,
              const uint32_t* __restrict__ sorted, XYZZ<Fq2_28>* __restrict__ buckets, uint32_t total_buckets,
              uint32_t heavy_thr, uint32_t* __restrict__ redo) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = gt >> 1, comp = gt & 1u;
  if (t >= total_buckets) return;  // pair-uniform
  const uint32_t b = perm[t];
  const uint32_t cnt = count[b];
  if (cnt > heavy_thr) return;
  const uint32_t beg = begin[b], end = beg + cnt;
  Fq28* dst = reinterpret_cast<Fq28*>(buckets + b);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
  auto load_point = [&](uint32_t v, Affine<Fq2P>& p) {
    const Fq28* src = reinterpret_cast<const Fq28*>(bases + (v & 0x7fffffffu));  // x.c0 x.c1 y.c0 y.c1
    p.x.v = ld_comp(src + comp);
    p.y.v = ld_comp(src + 2 + comp);
    // infinity = all four components exact zeros: OR of this lane's words, combined with the partner's; the lowest
    // limb of y decides for all but 2^-56 of the finite entries, the full test runs behind that (pair-uniform) branch
    uint32_t o = (uint32_t)p.y.v.l[0];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    if (o != 0) return false;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) o |= (uint32_t)p.x.v.l[i] | (uint32_t)p.y.v.l[i];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    return o == 0;  // (the digit's sign is applied by the caller: first point, or folded into madd_generic)
  };
  XYZZ<Fq2P> acc;
  uint32_t j = beg;
  for (;; j++) {
    if (j >= end) {
      const Fq28 z = Fq28::zero();
      for (int k = 0; k < 4; k++) st_comp(dst + 2 * k + comp, z);
      return;
    }
    Affine<Fq2P> p;
    const uint32_t v = sorted[j];
    if (load_point(v, p)) continue;
    acc.x = p.x;
    acc.y = (v >> 31) ? p.y.neg() : p.y;
    acc.zz = Fq2P::one();
    acc.zzz = Fq2P::one();
    j++;
    break;
  }
  for (; j < end; j++) {
    Affine<Fq2P> p;
    const uint32_t v = sorted[j];
    if (load_point(v, p)) continue;
    if (!madd_generic(acc, p, 0u - (v >> 31))) {  // pair-uniform (Fq2P::is_zero exchanges the halves)
      if (comp == 0) redo[1 + atomicAdd(redo, 1u)] = b;
      return;
    }
  }
  st_comp(dst + comp, acc.x.v);
  st_comp(dst + 2 + comp, acc.y.v);
  st_comp(dst + 4 + comp, acc.zz.v);
  st_comp(dst + 6 + comp, acc.zzz.v);
}

// G2, TWO lane pairs per bucket (pair 0: entries beg, beg + 2, ...; pair 1: beg + 1, beg + 3, ...), pair 0 adds the two partial
// sums at the end: k_accum_g1_split2's idea for the G2 launch of ONE small proof, which is 0.3 ms of chip time but lasts as long
// as the fullest bucket's chain of dependent mixed additions (17 per bucket at 2^14: 0.65 ms, the longest link of a 1.9 ms
// proof).  One wave per SIMD (the general addition on top of the loop's state does not fit 256 registers); only for plans of
// <= 2^16 buckets on an otherwise idle chip.  A doubling in either half or in the final addition sends the bucket to the redo list.
template <int UNUSED = 0>
__global__ void __launch_bounds__(64, 1)
k_accum_g2_split2(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ begin, const uint32_t* __restrict__ count,
                  const uint32_t* __restrict__ perm, const uint32_t* __restrict__ sorted, XYZZ<Fq2_28>* __restrict__ buckets,
                  uint32_t total_buckets, uint32_t heavy_thr, uint32_t* __restrict__ redo) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = gt >> 2, sub = (gt >> 1) & 1u, comp = gt & 1u;
  if (t >= total_buckets) return;  // quad-uniform
  const uint32_t b = perm[t];
  const uint32_t cnt = count[b];
  if (cnt > heavy_thr) return;
  const uint32_t beg = begin[b], end = beg + cnt;
  Fq28* dst = reinterpret_cast<Fq28*>(buckets + b);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
  auto load_point = [&](uint32_t v, Affine<Fq2P>& p) {  // true = the point at infinity (see k_accum_g2_nc)
    const Fq28* src = reinterpret_cast<const Fq28*>(bases + (v & 0x7fffffffu));
    p.x.v = ld_comp(src + comp);
    p.y.v = ld_comp(src + 2 + comp);
    uint32_t o = (uint32_t)p.y.v.l[0];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    if (o != 0) return false;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) o |= (uint32_t)p.x.v.l[i] | (uint32_t)p.y.v.l[i];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    return o == 0;
  };
  XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
  bool has = false, bad = false;
  uint32_t j = beg + sub;
  for (; j < end; j += 2) {  // this pair's first entry that is not the point at infinity starts its sum
    Affine<Fq2P> p;
    const uint32_t v = sorted[j];
    if (load_point(v, p)) continue;
    acc.x = p.x;
    acc.y = (v >> 31) ? p.y.neg() : p.y;
    acc.zz = Fq2P::one();
    acc.zzz = Fq2P::one();
    has = true;
    j += 2;
    break;
  }
  for (; j < end; j += 2) {
    Affine<Fq2P> p;
    const uint32_t v = sorted[j];
    if (load_point(v, p)) continue;
    if (!madd_generic(acc, p, 0u - (v >> 31))) {  // pair-uniform
      bad = true;
      break;
    }
  }
  // the other pair's state and partial sum (lane ^ 2: the same component of the other pair)
  const uint32_t flags = (has ? 1u : 0u) | (bad ? 2u : 0u);
  const uint32_t pflags = (uint32_t)__shfl_xor((int)flags, 2);
  XYZZ<Fq2P> o;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) {
    o.x.v.l[i] = __shfl_xor(acc.x.v.l[i], 2);
    o.y.v.l[i] = __shfl_xor(acc.y.v.l[i], 2);
    o.zz.v.l[i] = __shfl_xor(acc.zz.v.l[i], 2);
    o.zzz.v.l[i] = __shfl_xor(acc.zzz.v.l[i], 2);
  }
  if (sub != 0) return;
  bool redo_it = ((flags | pflags) & 2u) != 0;
  if (!redo_it) {
    if (has && (pflags & 1u)) redo_it = !add_generic(acc, o);  // pair-uniform
    else if (!has) acc = (pflags & 1u) ? o : XYZZ<Fq2P>::infinity();
  }
  if (redo_it) {
    if (comp == 0) redo[1 + atomicAdd(redo, 1u)] = b;  // k_accum_redo recomputes the bucket with the complete addition
  } else {
    st_comp(dst + comp, acc.x.v);
    st_comp(dst + 2 + comp, acc.y.v);
    st_comp(dst + 4 + comp, acc.zz.v);
    st_comp(dst + 6 + comp, acc.zzz.v);
  }
}

// the listed buckets again, with the complete addition (rare: repeated bases with equal digits)
// The kernel also CLEARS the list for the slot's next MSM: every workgroup reads the length first, and the last one to
// finish (a ticket word behind the list) resets length and ticket -- no hipMemsetAsync launch in front of an accumulation
// (a kernel of its own that waited up to 0.8 ms for a slot on a full chip).  The buffer is zeroed once when it is allocated.
// Round 5: one WAVE per listed bucket (lane l adds entries l, l + 64, ...; LDS tree of the 64 partial sums) instead of one
// lane: the synthetic bases of the benchmarks and tests, P_i = G + i Q, meet P + P by arithmetic coincidence (a running sum
// P_a - P_b + P_c IS P_(a - b + c)) about once per 10^7 insertions, and the one lane that then walked its bucket's ~32
// entries with 16 us per dependent addition held the whole reduction back by 0.2 - 0.5 ms per MSM.
template <class F>
__device__ __forceinline__ XYZZ<F> block_tree_sum(XYZZ<F> acc, XYZZ<F>* sh);
template <class F>
__global__ void __launch_bounds__(64)
k_accum_redo(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin,
             const uint32_t* __restrict__ count, const uint32_t* __restrict__ sorted,
             XYZZ<F>* __restrict__ buckets, uint32_t* __restrict__ redo, uint32_t* __restrict__ ticket, uint32_t into) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds_raw);
  const uint32_t n = redo[0];
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {  // block-uniform
    const uint32_t b = redo[1 + k];
    const uint32_t beg = begin[b], end = beg + count[b];
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (uint32_t j = beg + threadIdx.x; j < end; j += blockDim.x) {
      const uint32_t v = sorted[j];
      Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
      if (v >> 31) p.y = p.y.neg();
      acc.madd(p);
    }
    acc = block_tree_sum(acc, sh);
    if (threadIdx.x == 0) {
      // into: the accumulation was adding to an earlier MSM's bucket sums and left the listed buckets untouched
      if (into) acc.add(load_vec(buckets + b));
      store_vec(buckets + b, acc);
    }
    __syncthreads();
  }
  // every workgroup has read the length by now; the last one to get here clears the list for the slot's next MSM
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    redo[0] = 0;
    *ticket = 0;
  }
}

// Heavy buckets (repeated scalars, booleans: one bucket can hold 20 % of all points):
// MSM_HSPLIT workgroups share one bucket, each reduces a sub-range with an LDS tree into
// heavy_partial[h][r]; the last of them to finish adds the partials.  Buckets beyond the
// partial-slot capacity (pathological inputs) fall back to one workgroup per bucket.
constexpr uint32_t MSM_HSPLIT = 64;
constexpr uint32_t MSM_HEAVY_CAP = 1024;

template <class F>
__device__ __forceinline__ XYZZ<F> block_tree_sum(XYZZ<F> acc, XYZZ<F>* sh) {
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      acc.add(sh[threadIdx.x + s]);
      sh[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// Round 5: the heavy buckets' additions in a kernel that RUNS BESIDE the light accumulation.
// k_accum_heavy (below) adds with the complete group law: 320 registers and scratch, one wave per SIMD (1.6 G additions/s
// against the light kernel's 6.5), and -- two-wave workgroups of that size -- it is not placed while the light kernel
// still has workgroups to issue: the heavy buckets of a witness-like MSM started when the light ones
// had finished.  Now the additions of POINTS into partial sums -- all but a few per cent of the work -- are made by
// one-wave workgroups with the light kernel's own loop (call-free mixed additions, next table entry prefetched through
// LDS: 168 registers, no scratch), every lane on its own stride of a sub-range:
//   k_heavy_plan       one workgroup: list entry h -> (bucket, first entry, points, first wave-item); points per lane
//                      chosen so that the partial sums fit the pool
//   k_accum_heavy_nc   wave-item = 64 x PL consecutive entries of a heavy bucket; lane l adds entries l, l + 64, ... and
//                      stores ONE partial sum (G2: 32 lane pairs); a lane that meets P + P stores a marker instead
//   k_accum_heavy<..>  in PARTIAL mode: sums each bucket's partials with the complete law (LDS trees, ticket for split
//                      buckets: the code that used to add the points); a marked partial is recomputed there, serially,
//                      from its <= PL points.  In POINT mode (plan too long for the pool, partitioned big windows,
//                      A/B accumulate-into forms) it is the kernel of rounds 1-4.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MSM_HNC_ENT = 1024;        // list entries the plan holds (= k_heavy_plan's block, <= MSM_HEAVY_CAP ticket words)
constexpr uint32_t MSM_HNC_POOL = 131072;     // first-level partial sums per MSM slot
constexpr uint32_t MSM_HNC_MIN_PL = 8;        // points per lane at least (the tree behind costs ~1.4 additions per partial)
constexpr uint32_t MSM_HPLAN_WORDS = 4 + 4 * (MSM_HNC_ENT + 1);
// hplan: [0] wave-items, [1] points per lane PL, [2] entries, [3] mode (0 = partial sums by k_accum_heavy_nc, 1 = point mode),
// then per entry {bucket, first slot in sorted[], points, first wave-item}; entry [entries] = {.., .., .., wave-items}
// ONE WAVE (16 entries per lane): a 1 024-thread workgroup is not placed while accumulation waves hold 504 of a SIMD's 512
// registers (it sat 0.66 - 2.2 ms in front of the heavy kernels in the first traces of this round); a wave fits the slot
// one retiring accumulation wave frees.
template <int UNUSED = 0>
__global__ void __launch_bounds__(64)
k_heavy_plan(const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ count, const uint32_t* __restrict__ begin,
             uint32_t* __restrict__ hplan, uint32_t lanes, uint32_t force_point_mode) {
  constexpr uint32_t PER = MSM_HNC_ENT / 64;
  const uint32_t lane = threadIdx.x, n = heavy[0];
  if (force_point_mode || n > MSM_HNC_ENT) {
    if (lane == 0) {
      hplan[0] = 0;
      hplan[1] = 0;
      hplan[2] = 0;
      hplan[3] = 1;
    }
    return;
  }
  // lane owns entries [lane * PER, (lane + 1) * PER)
  uint32_t c[PER], mine = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t h = lane * PER + k;
    c[k] = h < n ? count[heavy[1 + h]] : 0u;
    mine += c[k];
  }
  auto wave_incl = [&](uint32_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)v, off);
      if ((int)lane >= off) v += t;
    }
    return v;
  };
  const uint32_t total = (uint32_t)__shfl((int)wave_incl(mine), 63);  // (entries < 2^32: the sort's slot arithmetic)
  // partial sums = lanes x wave-items <= total / PL + lanes x n
  const uint32_t room = MSM_HNC_POOL - lanes * MSM_HNC_ENT;
  uint32_t pl = (total + room - 1) / room;
  if (pl < MSM_HNC_MIN_PL) pl = MSM_HNC_MIN_PL;
  const uint32_t per_item = lanes * pl;
  uint32_t items[PER], my_items = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    items[k] = (c[k] + per_item - 1) / per_item;
    my_items += items[k];
  }
  const uint32_t incl = wave_incl(my_items);
  uint32_t run = incl - my_items;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t h = lane * PER + k;
    if (h < n) {
      const uint32_t b = heavy[1 + h];
      uint32_t* e = hplan + 4 + 4 * h;
      e[0] = b;
      e[1] = begin[b];
      e[2] = c[k];
      e[3] = run;
    }
    run += items[k];
  }
  if (lane == 63) {
    hplan[0] = incl;
    hplan[1] = pl;
    hplan[2] = n;
    hplan[3] = 0;
    hplan[4 + 4 * n + 3] = incl;
  }
}
// the plan entry that owns wave-item `it` (last e with first_item[e] <= it): uniform over the wave
__device__ __forceinline__ uint32_t heavy_plan_entry(const uint32_t* __restrict__ ent, uint32_t n_ent, uint32_t it) {
  uint32_t lo = 0, hi = n_ent;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ent[4 * mid + 3] <= it) lo = mid;
    else hi = mid;
  }
  return lo;
}
// what a lane that met P + P (or P - P) leaves instead of its partial sum: "infinity" with a non-zero zzz word
template <class F>
__device__ __forceinline__ XYZZ<F> heavy_marker() {
  XYZZ<F> m = XYZZ<F>::infinity();
  m.zzz.l[0] = 1;
  return m;
}

template <class F, int W>
__global__ void __launch_bounds__(64, W)
k_accum_heavy_nc(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ hplan,
                 XYZZ<F>* __restrict__ pool) {
  constexpr int CHUNKS = sizeof(Affine<F>) / 16;
  __shared__ uint4 tile[CHUNKS][64];
  if (hplan[3] != 0) return;  // point mode: k_accum_heavy takes the list
  const uint32_t n_items = hplan[0], pl = hplan[1], n_ent = hplan[2];
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t lane = threadIdx.x;
  auto fetch = [&](uint32_t v) {
    const char* src = reinterpret_cast<const char*>(bases + (v & 0x7fffffffu));
#pragma unroll
    for (int q = 0; q < CHUNKS; q++) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 16 * q), (lds_ptr_t)&tile[q][0], 16, 0, 0);
  };
  auto take = [&](Affine<F>& p) {
    uint4* d = reinterpret_cast<uint4*>(&p);
#pragma unroll
    for (int q = 0; q < CHUNKS; q++) d[q] = tile[q][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const uint32_t e = heavy_plan_entry(ent, n_ent, it);
    const uint32_t beg = ent[4 * e + 1], cnt = ent[4 * e + 2], first = ent[4 * e + 3];
    const uint32_t w0 = beg + (it - first) * 64u * pl;
    const uint32_t w1 = (w0 + 64u * pl < beg + cnt) ? w0 + 64u * pl : beg + cnt;
    XYZZ<F>* const dst = pool + (size_t)it * 64 + lane;
    XYZZ<F> acc;
    uint32_t j = w0 + lane;
    bool have = false;
    for (; j < w1; j += 64) {  // the lane's first entry that is not the point at infinity starts its sum
      const uint32_t v = sorted[j];
      Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
      if (affine_is_zero_words(p)) continue;
      if (v >> 31) p.y = p.y.neg();
      acc.x = p.x;
      acc.y = p.y;
      acc.zz = F::one();
      acc.zzz = F::one();
      have = true;
      j += 64;
      break;
    }
    if (!have) {
      store_vec(dst, XYZZ<F>::infinity());
      continue;
    }
    uint32_t v_cur = 0, v_next = 0;
    if (j < w1) {
      v_cur = sorted[j];
      fetch(v_cur);
      if (j + 64 < w1) v_next = sorted[j + 64];
    }
    bool bad = false;
    for (; j < w1; j += 64) {
      Affine<F> p;
      take(p);
      const uint32_t v = v_cur;
      if (j + 64 < w1) {
        fetch(v_next);
        v_cur = v_next;
        if (j + 128 < w1) v_next = sorted[j + 128];
      }
      if (affine_is_zero_words(p)) continue;
      if (!madd_generic(acc, p, 0u - (v >> 31))) {
        bad = true;  // k_accum_heavy recomputes this lane's <= PL points with the complete law
        break;
      }
    }
    if (bad) store_vec(dst, heavy_marker<F>());
    else store_vec(dst, acc);
    // (a lane that left the loop early still has a direct-to-LDS load in flight into its own column of the tile: the next
    // item's first fetch into the same column is ordered behind it, and take() waits for both)
  }
}

// G2: 32 lane pairs per wave-item (field28.hpp Fq2P), the light G2 kernel's loop
template <int W>
__global__ void __launch_bounds__(64, W)
k_accum_heavy_nc_g2(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ hplan,
                    XYZZ<Fq2_28>* __restrict__ pool) {
  if (hplan[3] != 0) return;
  const uint32_t n_items = hplan[0], pl = hplan[1], n_ent = hplan[2];
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t pair = threadIdx.x >> 1, comp = threadIdx.x & 1u;
  auto load_point = [&](uint32_t v, Affine<Fq2P>& p) {  // true = the point at infinity (all four components exact zeros)
    const Fq28* src = reinterpret_cast<const Fq28*>(bases + (v & 0x7fffffffu));
    p.x.v = ld_comp(src + comp);
    p.y.v = ld_comp(src + 2 + comp);
    uint32_t o = (uint32_t)p.y.v.l[0];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    if (o != 0) return false;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) o |= (uint32_t)p.x.v.l[i] | (uint32_t)p.y.v.l[i];
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    return o == 0;
  };
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
    const uint32_t e = heavy_plan_entry(ent, n_ent, it);
    const uint32_t beg = ent[4 * e + 1], cnt = ent[4 * e + 2], first = ent[4 * e + 3];
    const uint32_t w0 = beg + (it - first) * 32u * pl;
    const uint32_t w1 = (w0 + 32u * pl < beg + cnt) ? w0 + 32u * pl : beg + cnt;
    Fq28* const dst = reinterpret_cast<Fq28*>(pool + (size_t)it * 32 + pair);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
    XYZZ<Fq2P> acc;
    uint32_t j = w0 + pair;
    bool have = false, bad = false;
    for (; j < w1; j += 32) {
      Affine<Fq2P> p;
      const uint32_t v = sorted[j];
      if (load_point(v, p)) continue;
      acc.x = p.x;
      acc.y = (v >> 31) ? p.y.neg() : p.y;
      acc.zz = Fq2P::one();
      acc.zzz = Fq2P::one();
      have = true;
      j += 32;
      break;
    }
    if (have) {
      for (; j < w1; j += 32) {
        Affine<Fq2P> p;
        const uint32_t v = sorted[j];
        if (load_point(v, p)) continue;
        if (!madd_generic(acc, p, 0u - (v >> 31))) {  // pair-uniform
          bad = true;
          break;
        }
      }
    }
    if (!have || bad) {
      const Fq28 z = Fq28::zero();
      Fq28 mk = z;
      mk.l[0] = (bad && comp == 0) ? 1 : 0;  // marker: zzz.c0 word 0 (k_accum_heavy_g2_split recomputes the pair's points)
      st_comp(dst + comp, z);
      st_comp(dst + 2 + comp, z);
      st_comp(dst + 4 + comp, z);
      st_comp(dst + 6 + comp, mk);
    } else {
      st_comp(dst + comp, acc.x.v);
      st_comp(dst + 2 + comp, acc.y.v);
      st_comp(dst + 4 + comp, acc.zz.v);
      st_comp(dst + 6 + comp, acc.zzz.v);
    }
  }
}

// grid = (MSM_HSPLIT, groups).  The sub-ranges of a split bucket are added up by whichever of its workgroups finishes
// last (ticket word per heavy bucket, left at zero for the slot's next MSM): there is no separate combine launch -- a
// kernel of 330-register waves (G2) that found no SIMD until the accumulation beside it had drained, with the reduction
// waiting behind it (0.23 ms of one 2^14 proof's critical path for a list in which no bucket was split at all).
//
// Round 5: a bucket is cut into as many sub-ranges as it can feed (~4 points per thread), up to MSM_HSPLIT_MAX -- no longer
// at most 64.  A witness of bits puts 630 000 points into ONE bucket per query; 64 sub-ranges x 64 lane pairs left every
// lane pair of the G2 kernel a chain of 154 dependent additions on an otherwise empty chip: 5.3 ms of a 9.5 ms proof
// (profiles/r05/experiments/heavy_bucket_before.txt).  The partial sums live in one pool per MSM slot (MSM_HEAVY_CAP x
// MSM_HSPLIT entries as before), handed out in list order: entry h owns [item0, item0 + nsplit) where item0 = the
// sub-ranges of the entries before it -- the number every workgroup already computes to deal the work items.
constexpr uint32_t MSM_HSPLIT_MAX = 1024;
constexpr uint32_t MSM_HPOOL = MSM_HEAVY_CAP * MSM_HSPLIT;  // partial-sum slots per MSM slot
// sub-ranges of list entry h (count points, pts points per workgroup pass), given the pool slots already handed out
__device__ __forceinline__ uint32_t heavy_nsplit(uint32_t h, uint32_t count, uint32_t pts, uint32_t item0) {
  if (h >= MSM_HEAVY_CAP) return 1u;  // beyond the ticket words: one workgroup per bucket
  uint32_t n = (count + pts - 1) / pts;
  n = n < 1 ? 1 : (n > MSM_HSPLIT_MAX ? MSM_HSPLIT_MAX : n);
  if (item0 + n > MSM_HPOOL) n = 1;  // pool exhausted (thousands of split buckets: pathological): unsplit
  return n;
}
template <class F>
__global__ void __launch_bounds__(MSM_TREE_T)
k_accum_heavy(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin,
              const uint32_t* __restrict__ count, const uint32_t* __restrict__ heavy,
              const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets,
              XYZZ<F>* __restrict__ heavy_partial, uint32_t* __restrict__ ticket, uint32_t into, uint32_t h_first,
              uint32_t h_limit, const uint32_t* __restrict__ hplan, const XYZZ<F>* __restrict__ pool_nc) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds_raw);
  __shared__ uint32_t is_last;
  // PARTIAL mode (hplan given and its mode word 0): the "points" of list entry h are the partial sums k_accum_heavy_nc left
  // in pool_nc -- 64 per wave-item, in wave-item order -- and the list is the plan's; POINT mode: the kernel of rounds 1-4
  const bool pmode = hplan != nullptr && hplan[3] == 0;
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t pl_nc = pmode ? hplan[1] : 0u;
  // a partial sum, or -- where its lane met P + P -- the lane's points again with the complete law
  auto partial_at = [&](uint32_t h, uint32_t j) {
    XYZZ<F> v = load_vec(pool_nc + j);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F::NL; i++) o |= (uint32_t)v.zz.l[i];
    if (o == 0 && v.zzz.l[0] == 1) {
      const uint32_t it = j >> 6, ln = j & 63u;
      const uint32_t beg = ent[4 * h + 1], cnt = ent[4 * h + 2], first = ent[4 * h + 3];
      const uint32_t w0 = beg + (it - first) * 64u * pl_nc;
      const uint32_t w1 = (w0 + 64u * pl_nc < beg + cnt) ? w0 + 64u * pl_nc : beg + cnt;
      v = XYZZ<F>::infinity();
      for (uint32_t jj = w0 + ln; jj < w1; jj += 64) {
        const uint32_t sv = sorted[jj];
        Affine<F> p = load_vec(bases + (sv & 0x7fffffffu));
        if (sv >> 31) p.y = p.y.neg();
        v.madd(p);
      }
    }
    return v;
  };
  // list entries [h_first, min(length, h_limit)): one launch takes the whole list, or -- plans whose partial top window
  // puts thousands of buckets above the threshold (windowed c = 20: 2^14 buckets of 4 500 points at 2^26 terms) -- the
  // first MSM_HEAVY_CAP entries go to a (MSM_HSPLIT, 8) grid and the rest to a second launch with one workgroup per
  // list entry (a (MSM_HSPLIT, 8) grid would walk them with 8 workgroups)
  const uint32_t n_heavy = pmode ? hplan[2] : (heavy[0] < h_limit ? heavy[0] : h_limit);
  // into: the bucket's sum is added to what an earlier MSM left in the bucket (complete addition, one thread)
  auto put = [&](uint32_t b, XYZZ<F> v) {
    if (into) v.add(load_vec(buckets + b));
    store_vec(buckets + b, v);
  };
  // Work items = (bucket, sub-range) pairs, numbered through the list and dealt round-robin to ALL workgroups of the grid.
  // (Rounds 1-3 gave grid row y the buckets y, y + 8, ... and column x the sub-range x: the 64 buckets of a group of small
  // proofs -- one per proof, ~600 points = 2 sub-ranges each -- were walked 8 at a time by 2 of the 64 columns: 5.8 ms per
  // launch at 2^14, a quarter of the group's kernel time; dealt flat the same list is one round of 128 workgroups.)
  const uint32_t n_wg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
  uint32_t item0 = 0;  // number of the bucket's first work item = its first pool slot
  // (a launch for the entries beyond the split capacity -- one item per bucket -- strides the list directly)
  const bool tail = h_first >= MSM_HEAVY_CAP;
  for (uint32_t h = tail ? h_first + wg : h_first; h < n_heavy; h += tail ? n_wg : 1u) {
    const uint32_t b = pmode ? ent[4 * h] : heavy[1 + h];
    // (partial mode: positions in pool_nc -- 64 partial sums per wave-item of the entry)
    const uint32_t cnt = pmode ? (ent[4 * (h + 1) + 3] - ent[4 * h + 3]) * 64u : count[b];
    const uint32_t beg0 = pmode ? ent[4 * h + 3] * 64u : begin[b];
    // as many sub-ranges as the bucket can feed: ~4 points per thread before the tree (a bucket of 400 points on
    // all 64 x 128 threads is 64 trees of points at infinity: measured 10 % of all instructions of a 2^14 group).
    // (8 points per thread -- one workgroup, no partials and no second tree for the 600 bit variables of a small
    // proof -- measured no better: 2^14 2 501 against 2 554-2 573 proofs/s.)
    const uint32_t nsplit = tail ? 1u : heavy_nsplit(h, cnt, 4 * MSM_TREE_T, item0);
    const uint32_t base = item0;
    // this workgroup's sub-ranges of the bucket: item numbers base + r = wg (mod n_wg)
    const uint32_t r0 = tail ? 0u : (wg + n_wg - base % n_wg) % n_wg;
    item0 += nsplit;
    for (uint32_t r = r0; r < nsplit; r += n_wg) {  // block-uniform
      uint32_t beg = beg0, end = beg0 + cnt;
      if (nsplit > 1) {
        const uint32_t len = (cnt + nsplit - 1) / nsplit;
        const uint32_t sb = beg + r * len;
        end = (sb + len < end) ? sb + len : end;
        beg = sb < end ? sb : end;
      }
      XYZZ<F> acc = XYZZ<F>::infinity();
      if (pmode) {
        for (uint32_t j = beg + threadIdx.x; j < end; j += blockDim.x) acc.add(partial_at(h, j));
      } else {
        for (uint32_t j = beg + threadIdx.x; j < end; j += blockDim.x) {
          const uint32_t v = sorted[j];
          Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
          if (v >> 31) p.y = p.y.neg();
          acc.madd(p);
        }
      }
      acc = block_tree_sum(acc, sh);
      if (nsplit == 1) {
        if (threadIdx.x == 0) put(b, acc);
      } else {
        if (threadIdx.x == 0) {
          store_vec(heavy_partial + (size_t)base + r, acc);
          __threadfence();  // the partial is visible device-wide before the ticket is taken
          is_last = atomicAdd(ticket + h, 1u) == nsplit - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (is_last) {  // block-uniform
          __threadfence();
          XYZZ<F> v = XYZZ<F>::infinity();
          for (uint32_t k = threadIdx.x; k < nsplit; k += blockDim.x) v.add(load_vec(heavy_partial + (size_t)base + k));
          v = block_tree_sum(v, sh);
          if (threadIdx.x == 0) {
            put(b, v);
            ticket[h] = 0;
          }
        }
      }
      __syncthreads();
    }
  }
}

// thread t handles buckets [t*SEG, (t+1)*SEG) of one window (global segment id)
// buckets2 (optional): the bucket array of a SECOND MSM over the same bucket set whose result is only ever added to this
// one's (the prover's L and H queries: C = ... + L + H).  Its buckets join the running sum here -- three additions per bucket
// for the pair instead of two each, and one tree sum / host combine instead of two -- and nothing else of the two MSMs
// knows about the other: both accumulate into their own arrays with the ordinary kernels.
template <class F>
__global__ void __launch_bounds__(256)
k_segreduce(const XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ segsum, XYZZ<F>* __restrict__ segw,
            uint32_t total_segs, int seg, const XYZZ<F>* __restrict__ buckets2, const XYZZ<F>* __restrict__ buckets3) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_segs) return;
  XYZZ<F> run = XYZZ<F>::infinity();
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int i = seg - 1; i >= 0; i--) {
    XYZZ<F> bk = load_vec(buckets + (size_t)t * seg + i);
    run.add(bk);
    if (buckets2) {
      bk = load_vec(buckets2 + (size_t)t * seg + i);
      run.add(bk);
    }
    if (buckets3) {
      bk = load_vec(buckets3 + (size_t)t * seg + i);
      run.add(bk);
    }
    acc.add(run);
  }
  store_vec(segsum + t, run);
  store_vec(segw + t, acc);
}

// The same kernel at two waves per SIMD (256 registers, 656 B of scratch instead of 464): for the reductions that have the chip to
// themselves and enough waves to use the room -- the 13 x 2^19 buckets of the big windowed plans, 5.66 -> 4.98 ms at 2^26 terms.
// (The prover's and the small plans' reductions keep the form above: they run beside accumulations or are latency chains.)
template <class F>
__global__ void __launch_bounds__(256, 2)
k_segreduce_w2(const XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ segsum, XYZZ<F>* __restrict__ segw,
            uint32_t total_segs, int seg, const XYZZ<F>* __restrict__ buckets2, const XYZZ<F>* __restrict__ buckets3) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_segs) return;
  XYZZ<F> run = XYZZ<F>::infinity();
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int i = seg - 1; i >= 0; i--) {
    XYZZ<F> bk = load_vec(buckets + (size_t)t * seg + i);
    run.add(bk);
    if (buckets2) {
      bk = load_vec(buckets2 + (size_t)t * seg + i);
      run.add(bk);
    }
    if (buckets3) {
      bk = load_vec(buckets3 + (size_t)t * seg + i);
      run.add(bk);
    }
    acc.add(run);
  }
  store_vec(segsum + t, run);
  store_vec(segw + t, acc);
}

// lane-pair split forms of the two reduction kernels for G2 (see k_accum_g2_split)
__device__ __forceinline__ XYZZ<Fq2P> ld_xyzz_split(const XYZZ<Fq2_28>* p, uint32_t comp) {
  const Fq28* s = reinterpret_cast<const Fq28*>(p);
  XYZZ<Fq2P> r;
  r.x.v = ld_comp(s + comp);
  r.y.v = ld_comp(s + 2 + comp);
  r.zz.v = ld_comp(s + 4 + comp);
  r.zzz.v = ld_comp(s + 6 + comp);
  return r;
}
__device__ __forceinline__ void st_xyzz_split(XYZZ<Fq2_28>* p, const XYZZ<Fq2P>& v, uint32_t comp) {
  Fq28* d = reinterpret_cast<Fq28*>(p);
  st_comp(d + comp, v.x.v);
  st_comp(d + 2 + comp, v.y.v);
  st_comp(d + 4 + comp, v.zz.v);
  st_comp(d + 6 + comp, v.zzz.v);
}

template <int UNUSED = 0>
__global__ void __launch_bounds__(256, 2)
k_segreduce_g2_split(const XYZZ<Fq2_28>* __restrict__ buckets, XYZZ<Fq2_28>* __restrict__ segsum,
                     XYZZ<Fq2_28>* __restrict__ segw, uint32_t total_segs, int seg) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = gt >> 1, comp = gt & 1u;
  if (t >= total_segs) return;  // pair-uniform
  XYZZ<Fq2P> run = XYZZ<Fq2P>::infinity();
  XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
  for (int i = seg - 1; i >= 0; i--) {
    XYZZ<Fq2P> bk = ld_xyzz_split(buckets + (size_t)t * seg + i, comp);
    run.add(bk);
    acc.add(run);
  }
  st_xyzz_split(segsum + t, run, comp);
  st_xyzz_split(segw + t, acc, comp);
}

// grid = (njobs, nwin, nchunk).  job 0: sum_t segw[w][t]; job j>=1: sum_{t: bit (j-1)} segsum[w][t];
// job plain_job (shared-bucket mode only): sum_t segsum[w][t].
// nchunk = 1: the workgroup sums the job's whole list and writes the result in the host representation.
// nchunk > 1 (small plans, where the reduction is a latency chain and the chip is empty): workgroup z sums the z-th
// slice of the list into stage[(w * njobs + job) * nchunk + z]; k_treesum_final adds the slices.  A list of 2^13 segments
// then costs 2 + 7 + 5 dependent additions instead of 64 + 7.
template <class F>
__global__ void __launch_bounds__(MSM_TREE_T)
k_treesum(const XYZZ<F>* __restrict__ segsum, const XYZZ<F>* __restrict__ segw, uint32_t segs_per_win,
          XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial, int plain_job, XYZZ<F>* __restrict__ stage,
          XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial_host) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds_raw);
  const int job = blockIdx.x;
  const int w = blockIdx.y;
  const uint32_t nchunk = gridDim.z, z = blockIdx.z;
  const XYZZ<F>* src = (job == 0 ? segw : segsum) + (size_t)w * segs_per_win;
  const bool whole = job == 0 || job == plain_job;
  const uint32_t len = whole ? segs_per_win : segs_per_win / 2;
  const uint32_t per = (len + nchunk - 1) / nchunk;
  const uint32_t lo = z * per, hi = lo + per < len ? lo + per : len;
  XYZZ<F> acc = XYZZ<F>::infinity();
  if (whole) {
    for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
      XYZZ<F> v = load_vec(src + t);
      acc.add(v);
    }
  } else {
    // bit job: enumerate only the segments whose bit (job - 1) is set, so that no lane idles through an addition
    const uint32_t b = (uint32_t)(job - 1), lowmask = (1u << b) - 1u;
    for (uint32_t u = lo + threadIdx.x; u < hi; u += blockDim.x) {
      const uint32_t t = ((u & ~lowmask) << 1) | (1u << b) | (u & lowmask);
      XYZZ<F> v = load_vec(src + t);
      acc.add(v);
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      acc.add(sh[threadIdx.x + s]);
      sh[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (nchunk == 1) {
      XYZZ<typename HostFieldOf<F>::type> o = {fq_from_fq28(acc.x), fq_from_fq28(acc.y), fq_from_fq28(acc.zz),
                                               fq_from_fq28(acc.zzz)};
      // device copy (the RCCL exchange gathers it) and, directly, the pinned host slot the combining thread reads: no
      // device-to-host copy behind the reduction (it ran as a blit kernel, 0.3 ms of waiting for SIMDs per MSM)
      store_vec(partial + (size_t)w * gridDim.x + job, o);
      store_vec(partial_host + (size_t)w * gridDim.x + job, o);
    } else {
      store_vec(stage + ((size_t)w * gridDim.x + job) * nchunk + z, acc);
    }
  }
}

// grid = (njobs, nwin), nchunk <= blockDim.x (a power of two) <= MSM_TREE_T: adds the slices of one (window, job)
template <class F>
__global__ void __launch_bounds__(MSM_TREE_T)
k_treesum_final(const XYZZ<F>* __restrict__ stage, uint32_t nchunk, XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial,
                XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial_host) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds_raw);
  const size_t idx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  XYZZ<F> acc = XYZZ<F>::infinity();
  if (threadIdx.x < nchunk) acc = load_vec(stage + idx * nchunk + threadIdx.x);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      acc.add(sh[threadIdx.x + s]);
      sh[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    XYZZ<typename HostFieldOf<F>::type> o = {fq_from_fq28(acc.x), fq_from_fq28(acc.y), fq_from_fq28(acc.zz),
                                             fq_from_fq28(acc.zzz)};
    store_vec(partial + idx, o);
    store_vec(partial_host + idx, o);
  }
}

// Lane-pair forms of the two tree-sum kernels for G2 (see k_accum_g2_split): a pair owns one list element, so the
// unsplit form's 330-register additions (one wave per SIMD, 45-60 us each in a latency chain) become G1-sized ones.
// Used for the sliced (small-plan) case, where the tree sums ARE the latency of a proof; blockDim.x / 2 pairs per workgroup.
__device__ __forceinline__ void st_host_split(XYZZ<Fq2>* dst, const XYZZ<Fq2P>& v, uint32_t comp) {
  Fq* d = reinterpret_cast<Fq*>(dst);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
  d[comp] = fq_from_fq28(v.x.v);
  d[2 + comp] = fq_from_fq28(v.y.v);
  d[4 + comp] = fq_from_fq28(v.zz.v);
  d[6 + comp] = fq_from_fq28(v.zzz.v);
}
__device__ __forceinline__ XYZZ<Fq2P> pair_tree_sum(XYZZ<Fq2P> acc, XYZZ<Fq2_28>* sh, uint32_t pair, uint32_t npair, uint32_t comp) {
  st_xyzz_split(sh + pair, acc, comp);
  __syncthreads();
  for (uint32_t s = npair / 2; s > 0; s >>= 1) {
    if (pair < s) {  // pair-uniform
      XYZZ<Fq2P> o = ld_xyzz_split(sh + pair + s, comp);
      acc.add(o);
      st_xyzz_split(sh + pair, acc, comp);
    }
    __syncthreads();
  }
  return acc;
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(2 * MSM_TREE_T, 2)
k_treesum_g2_split(const XYZZ<Fq2_28>* __restrict__ segsum, const XYZZ<Fq2_28>* __restrict__ segw, uint32_t segs_per_win,
                   XYZZ<Fq2>* __restrict__ partial, int plain_job, XYZZ<Fq2_28>* __restrict__ stage, XYZZ<Fq2>* __restrict__ partial_host) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<Fq2_28>* sh = reinterpret_cast<XYZZ<Fq2_28>*>(lds_raw);
  const int job = blockIdx.x;
  const int w = blockIdx.y;
  const uint32_t nchunk = gridDim.z, z = blockIdx.z;
  const uint32_t pair = threadIdx.x >> 1, comp = threadIdx.x & 1u, npair = blockDim.x >> 1;
  const XYZZ<Fq2_28>* src = (job == 0 ? segw : segsum) + (size_t)w * segs_per_win;
  const bool whole = job == 0 || job == plain_job;
  const uint32_t len = whole ? segs_per_win : segs_per_win / 2;
  const uint32_t per = (len + nchunk - 1) / nchunk;
  const uint32_t lo = z * per, hi = lo + per < len ? lo + per : len;
  const uint32_t b = whole ? 0u : (uint32_t)(job - 1), lowmask = (1u << b) - 1u;
  XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
  for (uint32_t u = lo + pair; u < hi; u += npair) {
    const uint32_t t = whole ? u : (((u & ~lowmask) << 1) | (1u << b) | (u & lowmask));
    XYZZ<Fq2P> v = ld_xyzz_split(src + t, comp);
    acc.add(v);
  }
  acc = pair_tree_sum(acc, sh, pair, npair, comp);
  if (pair == 0) {
    if (nchunk == 1)
    {
      st_host_split(partial + (size_t)w * gridDim.x + job, acc, comp);
      st_host_split(partial_host + (size_t)w * gridDim.x + job, acc, comp);
    }
    else
      st_xyzz_split(stage + ((size_t)w * gridDim.x + job) * nchunk + z, acc, comp);
  }
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(2 * MSM_TREE_T, 2)
k_treesum_final_g2_split(const XYZZ<Fq2_28>* __restrict__ stage, uint32_t nchunk, XYZZ<Fq2>* __restrict__ partial,
                         XYZZ<Fq2>* __restrict__ partial_host) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<Fq2_28>* sh = reinterpret_cast<XYZZ<Fq2_28>*>(lds_raw);
  const size_t idx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t pair = threadIdx.x >> 1, comp = threadIdx.x & 1u, npair = blockDim.x >> 1;
  XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
  if (pair < nchunk) acc = ld_xyzz_split(stage + idx * nchunk + pair, comp);
  acc = pair_tree_sum(acc, sh, pair, npair, comp);
  if (pair == 0) {
    st_host_split(partial + idx, acc, comp);
    st_host_split(partial_host + idx, acc, comp);
  }
}

// Lane-pair forms of the heavy-bucket and redo kernels for G2: same logic as k_accum_heavy / k_accum_redo with a lane
// pair per point (the unsplit forms are 330-register kernels with 55-75 us per dependent addition: 0.8 ms for one
// 600-point bucket of a 2^14 proof, on the critical path of the G2 reduction; and 0.1 ms to place an EMPTY redo kernel).
__device__ __forceinline__ Affine<Fq2P> ld_affine_split(const Affine<Fq2_28>* bases, uint32_t v, uint32_t comp) {
  const Fq28* src = reinterpret_cast<const Fq28*>(bases + (v & 0x7fffffffu));  // x.c0 x.c1 y.c0 y.c1
  Affine<Fq2P> p;
  p.x.v = ld_comp(src + comp);
  p.y.v = ld_comp(src + 2 + comp);
  if (v >> 31) p.y = p.y.neg();
  return p;
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(MSM_TREE_T, 2)
k_accum_heavy_g2_split(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ begin,
                       const uint32_t* __restrict__ count, const uint32_t* __restrict__ heavy,
                       const uint32_t* __restrict__ sorted, XYZZ<Fq2_28>* __restrict__ buckets,
                       XYZZ<Fq2_28>* __restrict__ heavy_partial, uint32_t* __restrict__ ticket, uint32_t h_first,
                       uint32_t h_limit, const uint32_t* __restrict__ hplan, const XYZZ<Fq2_28>* __restrict__ pool_nc) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<Fq2_28>* sh = reinterpret_cast<XYZZ<Fq2_28>*>(lds_raw);
  __shared__ uint32_t is_last;
  const uint32_t pair = threadIdx.x >> 1, comp = threadIdx.x & 1u, npair = blockDim.x >> 1;
  // PARTIAL / POINT mode: see k_accum_heavy (here 32 partial sums per wave-item, one per lane pair of k_accum_heavy_nc_g2)
  const bool pmode = hplan != nullptr && hplan[3] == 0;
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t pl_nc = pmode ? hplan[1] : 0u;
  auto partial_at = [&](uint32_t h, uint32_t j) {  // pair-uniform
    XYZZ<Fq2P> v = ld_xyzz_split(pool_nc + j, comp);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) o |= (uint32_t)v.zz.v.l[i];
    uint32_t mk = (comp == 0 && v.zzz.v.l[0] == 1) ? 1u : 0u;
    o |= (uint32_t)__builtin_amdgcn_mov_dpp((int)o, 0xB1, 0xF, 0xF, true);
    mk |= (uint32_t)__builtin_amdgcn_mov_dpp((int)mk, 0xB1, 0xF, 0xF, true);
    if (o == 0 && mk == 1) {
      const uint32_t it = j >> 5, pr = j & 31u;
      const uint32_t beg = ent[4 * h + 1], cnt = ent[4 * h + 2], first = ent[4 * h + 3];
      const uint32_t w0 = beg + (it - first) * 32u * pl_nc;
      const uint32_t w1 = (w0 + 32u * pl_nc < beg + cnt) ? w0 + 32u * pl_nc : beg + cnt;
      v = XYZZ<Fq2P>::infinity();
      for (uint32_t jj = w0 + pr; jj < w1; jj += 32) v.madd(ld_affine_split(bases, sorted[jj], comp));
    } else if (o == 0) {
      v = XYZZ<Fq2P>::infinity();
    }
    return v;
  };
  const uint32_t n_heavy = pmode ? hplan[2] : (heavy[0] < h_limit ? heavy[0] : h_limit);  // list range [h_first, h_limit): see k_accum_heavy
  const uint32_t n_wg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;  // flat deal of (bucket, sub-range) items
  uint32_t item0 = 0;
  const bool tail = h_first >= MSM_HEAVY_CAP;
  for (uint32_t h = tail ? h_first + wg : h_first; h < n_heavy; h += tail ? n_wg : 1u) {
    const uint32_t b = pmode ? ent[4 * h] : heavy[1 + h];
    const uint32_t cnt = pmode ? (ent[4 * (h + 1) + 3] - ent[4 * h + 3]) * 32u : count[b];
    const uint32_t beg0 = pmode ? ent[4 * h + 3] * 32u : begin[b];
    const uint32_t nsplit = tail ? 1u : heavy_nsplit(h, cnt, 4 * npair, item0);  // ~4 points per lane pair before the tree
    const uint32_t base = item0;  // first pool slot of the bucket's partial sums (see k_accum_heavy)
    const uint32_t r0 = tail ? 0u : (wg + n_wg - base % n_wg) % n_wg;
    item0 += nsplit;
    for (uint32_t r = r0; r < nsplit; r += n_wg) {  // block-uniform
      uint32_t beg = beg0, end = beg0 + cnt;
      if (nsplit > 1) {
        const uint32_t len = (cnt + nsplit - 1) / nsplit;
        const uint32_t sb = beg + r * len;
        end = (sb + len < end) ? sb + len : end;
        beg = sb < end ? sb : end;
      }
      XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
      if (pmode) {
        for (uint32_t j = beg + pair; j < end; j += npair) acc.add(partial_at(h, j));  // pair-uniform
      } else {
        for (uint32_t j = beg + pair; j < end; j += npair) acc.madd(ld_affine_split(bases, sorted[j], comp));  // pair-uniform
      }
      acc = pair_tree_sum(acc, sh, pair, npair, comp);
      if (nsplit == 1) {
        if (pair == 0) st_xyzz_split(buckets + b, acc, comp);
      } else {
        if (pair == 0) {
          st_xyzz_split(heavy_partial + (size_t)base + r, acc, comp);
          __threadfence();  // each lane's half of the partial is visible device-wide ...
        }
        __syncthreads();  // ... before the ticket is taken
        if (threadIdx.x == 0) is_last = atomicAdd(ticket + h, 1u) == nsplit - 1 ? 1u : 0u;
        __syncthreads();
        if (is_last) {  // block-uniform
          __threadfence();
          XYZZ<Fq2P> v = XYZZ<Fq2P>::infinity();
          for (uint32_t k = pair; k < nsplit; k += npair) v.add(ld_xyzz_split(heavy_partial + (size_t)base + k, comp));  // pair-uniform
          v = pair_tree_sum(v, sh, pair, npair, comp);
          if (pair == 0) st_xyzz_split(buckets + b, v, comp);
          if (threadIdx.x == 0) ticket[h] = 0;
        }
      }
      __syncthreads();
    }
  }
}
template <int UNUSED = 0>
__global__ void __launch_bounds__(64, 2)
k_accum_redo_g2_split(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ begin,
                      const uint32_t* __restrict__ count, const uint32_t* __restrict__ sorted,
                      XYZZ<Fq2_28>* __restrict__ buckets, uint32_t* __restrict__ redo, uint32_t* __restrict__ ticket) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  XYZZ<Fq2_28>* sh = reinterpret_cast<XYZZ<Fq2_28>*>(lds_raw);
  const uint32_t n = redo[0];
  const uint32_t pair = threadIdx.x >> 1, comp = threadIdx.x & 1u, npair = blockDim.x >> 1;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {  // block-uniform: a wave (32 lane pairs) per listed bucket
    const uint32_t b = redo[1 + k];
    const uint32_t beg = begin[b], end = beg + count[b];
    XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
    for (uint32_t j = beg + pair; j < end; j += npair) acc.madd(ld_affine_split(bases, sorted[j], comp));
    acc = pair_tree_sum(acc, sh, pair, npair, comp);
    if (pair == 0) st_xyzz_split(buckets + b, acc, comp);
    __syncthreads();
  }
  // every workgroup has read the length by now; the last one to get here clears the list for the slot's next MSM
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    redo[0] = 0;
    *ticket = 0;
  }
}

}  // namespace zkmi
#include "msm_quad.hpp"
namespace zkmi {

template <class F>
__global__ void __launch_bounds__(256)
k_bases_convert(const Affine<typename HostFieldOf<F>::type>* __restrict__ in, Affine<F>* __restrict__ out,
                uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<typename HostFieldOf<F>::type> p = load_vec(in + i);
  Affine<F> q = {fq28_from_fq(p.x), fq28_from_fq(p.y)};  // (0,0) stays exactly (0,0)
  store_vec(out + i, q);
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
template <class F>
void MsmEngine<F>::release() {
  if (buckets) (void)hipFree(buckets);
  if (segsum) (void)hipFree(segsum);
  if (segw) (void)hipFree(segw);
  if (partial) (void)hipFree(partial);
  if (tree_stage) (void)hipFree(tree_stage);
  tree_stage = nullptr;
  if (heavy_partial) (void)hipFree(heavy_partial);
  heavy_partial = nullptr;
  if (heavy_plan) (void)hipFree(heavy_plan);
  heavy_plan = nullptr;
  if (heavy_ticket) (void)hipFree(heavy_ticket);
  heavy_ticket = nullptr;
  if (redo) (void)hipFree(redo);
  redo = nullptr;
  if (h_partial) (void)hipHostFree(h_partial);
  // the events outlive a re-allocation: MsmSort::readers may still hold some of them (destroy_events() runs
  // from the destructor only)
  buckets = segsum = segw = nullptr;
  partial = h_partial = nullptr;
  cap_buckets = 0;
}

uint64_t msm_max_buckets(uint64_t n);

template <class F>
hipError_t MsmEngine<F>::reset_transients() {
  if (!redo || !heavy_ticket) return hipSuccess;
  hipError_t e = hipMemset(redo, 0, sizeof(uint32_t) * (cap_buckets + 2) * nslots);
  if (e != hipSuccess) return e;
  return hipMemset(heavy_ticket, 0, sizeof(uint32_t) * MSM_HEAVY_CAP * nslots);
}

template <class F>
void MsmEngine<F>::destroy_events() {
  for (int i = 0; i < SLOTS; i++) {
    hipEvent_t* evs[] = {&done[i], &acc_done[i], &pre[i], &heavy_done[i], &redo_done[i]};
    for (hipEvent_t* e : evs) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
  }
}

template <class F>
hipError_t MsmEngine<F>::ensure_slots(int n) {
  if (n > SLOTS) return hipErrorInvalidValue;
  if (n <= nslots) return hipSuccess;
  const bool sh = has_shared;
  const uint64_t mb = min_buckets;
  release();
  has_shared = sh;
  min_buckets = mb;
  nslots = n;
  return hipSuccess;
}

template <class F>
hipError_t MsmEngine<F>::reserve_buckets(uint64_t buckets) {
  if (buckets <= cap_buckets) return hipSuccess;
  const bool sh = has_shared;
  release();
  has_shared = sh;
  min_buckets = buckets;
  return reserve(1, sh);
}

template <class F>
hipError_t MsmEngine<F>::reserve(uint64_t n, bool shared_too) {
  uint64_t need = msm_max_buckets(n);
  if (need < min_buckets) need = min_buckets;
  const uint64_t forced = (uint64_t)(255 / 16 + 1) * (1u << 15);  // plan_override = 16 for any n
  if (need < forced) need = forced;
  shared_too = shared_too || has_shared;
  if (shared_too) {
    const MsmPlan sp = msm_make_plan_shared(n);  // shared-bucket plan of the same n
    const uint64_t sb = (uint64_t)sp.nwin * sp.nb;
    if (need < sb) need = sb;
  }
  has_shared = shared_too;
  if (need <= cap_buckets) return hipSuccess;
  release();
  hipError_t e;
  if ((e = hipMalloc(&buckets, sizeof(XYZZ<F>) * need * nslots)) != hipSuccess) return e;
  seg_cap = msm_max_segments(need);
  if ((e = hipMalloc(&segsum, sizeof(XYZZ<F>) * seg_cap * nslots)) != hipSuccess) return e;
  if ((e = hipMalloc(&segw, sizeof(XYZZ<F>) * seg_cap * nslots)) != hipSuccess) return e;
  if ((e = hipMalloc(&partial, sizeof(XYZZ<HF>) * SLOTS * SLOT_PTS)) != hipSuccess) return e;
  if ((e = hipMalloc(&tree_stage, sizeof(XYZZ<F>) * MSM_STAGE_PTS * nslots)) != hipSuccess) return e;
  // per slot: MSM_HPOOL partial sums of split buckets (k_accum_heavy) + MSM_HNC_POOL first-level ones (k_accum_heavy_nc)
  if ((e = hipMalloc(&heavy_partial, sizeof(XYZZ<F>) * ((size_t)MSM_HPOOL + MSM_HNC_POOL) * nslots)) != hipSuccess) return e;
  if ((e = hipMalloc(&heavy_plan, sizeof(uint32_t) * MSM_HPLAN_WORDS * nslots)) != hipSuccess) return e;
  if ((e = hipMalloc(&heavy_ticket, sizeof(uint32_t) * MSM_HEAVY_CAP * nslots)) != hipSuccess) return e;
  if ((e = hipMemset(heavy_ticket, 0, sizeof(uint32_t) * MSM_HEAVY_CAP * nslots)) != hipSuccess) return e;  // every use leaves zeros behind
  if ((e = hipMalloc(&redo, sizeof(uint32_t) * (need + 2) * nslots)) != hipSuccess) return e;  // one list per slot
  if ((e = hipMemset(redo, 0, sizeof(uint32_t) * (need + 2) * nslots)) != hipSuccess) return e;  // lengths and tickets start at zero
  if ((e = hipHostMalloc(&h_partial, sizeof(XYZZ<HF>) * SLOTS * SLOT_PTS, hipHostMallocCoherent)) != hipSuccess) return e;
  for (int i = 0; i < SLOTS; i++) {
    hipEvent_t* evs[] = {&done[i], &acc_done[i], &pre[i], &heavy_done[i], &redo_done[i]};
    for (hipEvent_t* ev : evs)
      if (!*ev && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return e;
  }
  cap_buckets = need;
  return hipSuccess;
}

static inline int msm_seg_bits(const MsmPlan& pl) {
  const uint32_t segs_per_win = pl.nb >> pl.seg_log;
  int seg_bits = 0;
  while ((1u << seg_bits) < segs_per_win) seg_bits++;
  return seg_bits;
}

template <class F>
hipError_t MsmEngine<F>::run_device(const MsmSort& sort, const Affine<F>* d_bases, hipStream_t st,
                                    hipStream_t st_reduce, PhaseTimer* prof, int ph_accum, int ph_reduce, int slot,
                                    hipStream_t st_heavy, int bucket_slot, int flags, int bucket_slot3) {
  const MsmSort* sp = &sort;
  return run_device_multi(&sp, &d_bases, 1, st, &st_reduce, prof, ph_accum, ph_reduce, &slot, st_heavy,
                          bucket_slot >= 0 ? &bucket_slot : nullptr, flags, bucket_slot3);
}

// nm MSMs of the same plan shape (sorts[m] may repeat: different tables over one digit sort): one fused accumulation
// launch where the kernel has a fused form (the call-free G1 kernels), otherwise nm launches in line; each MSM keeps
// its own slot and reduction stream.  All sorts must be complete on `st` (stream order or events) when this is called.
//
// Two MSMs whose results are only ever added (the prover's L and H queries: C = ... + L + H) can share one reduction: the
// first runs with MSM_RUN_NO_REDUCE (accumulation, heavy buckets and redo list only; its slot's `done` event is NOT recorded
// and it has no host result), the second names the first one's slot in bucket_slots[m] and
//   MSM_RUN_ADD_AT_REDUCE (the product): accumulates into its OWN array with the ordinary kernels; its k_segreduce adds both
//     arrays behind the first one's redo_done event -- the accumulations stay independent of each other;
//   without the flag (A/B library): its kernels add INTO the first one's array (k_accum_g1_nc<.., INTO>, k_accum_heavy /
//     k_accum_redo with into = 1) behind that event.
// Either way its reduction yields the sum of both.  Both sorts must plan the same bucket set (checked by the caller: the
// arrays are indexed by bucket id, and the ids mean the same digit values only under equal plans).
template <class F>
hipError_t MsmEngine<F>::run_device_multi(const MsmSort* const* sorts, const Affine<F>* const* d_bases, int nm, hipStream_t st,
                                          const hipStream_t* st_reduces, PhaseTimer* prof, int ph_accum, int ph_reduce,
                                          const int* slots, hipStream_t st_heavy, const int* bucket_slots, int flags, int third_slot) {
  if (nm < 1 || nm > MSM_MULTI_MAX) return hipErrorInvalidValue;
  // third_slot >= 0 (with MSM_RUN_ADD_AT_REDUCE, one MSM): a THIRD bucket array joins the segment sums
  if (third_slot >= nslots || (third_slot >= 0 && (nm != 1 || !(flags & MSM_RUN_ADD_AT_REDUCE) || !bucket_slots))) return hipErrorInvalidValue;
  const bool no_reduce = (flags & MSM_RUN_NO_REDUCE) != 0;
  // MSM_RUN_ADD_AT_REDUCE: bucket_slots[m] names a second SOURCE of this MSM's reduction (k_segreduce adds both arrays)
  // instead of the array its kernels add INTO
  const bool add_at_reduce = (flags & MSM_RUN_ADD_AT_REDUCE) != 0;
  bool any_into = false;
  for (int m = 0; m < nm; m++) {
    if (slots[m] < 0 || slots[m] >= nslots) return hipErrorInvalidValue;
    if (bucket_slots && (bucket_slots[m] < 0 || bucket_slots[m] >= nslots)) return hipErrorInvalidValue;
    any_into = any_into || (!add_at_reduce && bucket_slots && bucket_slots[m] != slots[m]);
  }
  if (add_at_reduce && std::is_same<F, Fq2_28>::value) return hipErrorInvalidValue;  // (the lane-pair segment sums have no second source)
  // the accumulate-into forms exist for the call-free G1 kernels, one MSM per launch
  if (any_into && (nm != 1 || std::is_same<F, Fq2_28>::value)) return hipErrorInvalidValue;
  const MsmPlan& pl = sorts[0]->plan;  // bucket count, windows, segment length: common to all (checked); heavy_thr is per sort
  for (int m = 1; m < nm; m++) {
    const MsmPlan& q = sorts[m]->plan;
    if (q.nwin != pl.nwin || q.nb != pl.nb || q.seg_log != pl.seg_log || q.shared != pl.shared || q.c != pl.c) return hipErrorInvalidValue;
  }
  const uint32_t tot_b = pl.nwin * pl.nb;
  const int T = 256;
  hipError_t e;
  // Heavy and light buckets are disjoint, so the heavy-bucket kernels only have to see the sort complete.  They run at the
  // head of the REDUCTION's stream.  With uniform scalars the heavy list is empty, but an empty 512-workgroup launch still
  // has to be placed: on a normal-priority stream it queued behind the next accumulation's workgroups (0.9 ms G1 / 9.8 ms
  // G2 average in the round-2 trace) and every reduction waited for it.  The reduction streams have high priority, so
  // there the launch is placed as soon as any workgroup retires, and no cross-stream event sits between it and the
  // reduction.  (A/B library, ZKMI_HEAVY_ON: 0 = in line on the accumulation stream; 1 = on `st_heavy`, round 2's side stream.)
  const int heavy_on = ZK_TUNE("ZKMI_HEAVY_ON", 2);
  // The product's accumulation kernels are the call-free ones at 3 (G1) / 2 (G2) waves per SIMD.  A/B
  // library: ZKMI_ACCUM / ZKMI_ACCUM_G2 = 0 | 1 select the retired generations (madd with an out-of-line doubling path),
  // 2 | 3 the call-free kernels at 2 / 3 waves; ZKMI_ACCUM_BLOCK = 64 | 256 threads per workgroup and ZKMI_ACCUM_ROUNDS = R
  // (every wave walks R load-ordered bucket groups) apply to the retired kernels.
  const int accum_mode = ZK_TUNE("ZKMI_ACCUM", 3);
  const int accum_mode_g2 = ZK_TUNE("ZKMI_ACCUM_G2", 2);
  const int mode = std::is_same<F, Fq2_28>::value ? accum_mode_g2 : accum_mode;
  const bool nocall = mode == 2 || mode == 3;
  // The reduction-side kernels with every complete addition split over a lane quad (msm_quad.hpp; G1 and BN254 G1): bit 0 =
  // segment sums, 1 = tree sums, 2 = redo pass, 3 = the summing of heavy-bucket partials.  One-wave workgroups at the
  // accumulation kernels' register count: placed beside them.  (A/B library: ZKMI_QUAD = mask; 0 = the one-lane kernels.)
  // The partitioned big windows keep the one-lane segment and tree sums: their reduction has the chip to itself.
  // Two contexts: ONE MSM or ONE proof by itself (host_spin: the latency path -- a chain of 4 products per dependent addition
  // instead of 14) and the batch prover's pipeline (throughput: the quad form issues 16 products for the 14 of the formula
  // plus its exchanges, 1.37 x the instructions of the one-lane addition).
  // G2 (the octet form, 8 lanes per point): ZKMI_QUAD_G2 / ZKMI_QUAD_G2_BATCH.
  constexpr bool g2_engine = std::is_same<F, Fq2_28>::value;
  const int quad_mask = g2_engine ? (host_spin ? ZK_TUNE("ZKMI_QUAD_G2", 15) : ZK_TUNE("ZKMI_QUAD_G2_BATCH", 12))
                                  : (host_spin ? ZK_TUNE("ZKMI_QUAD", 15) : ZK_TUNE("ZKMI_QUAD_BATCH", 12));
  using QPT = typename std::conditional<g2_engine, XYZZQ<Fq28, 0, true>, XYZZQ<F, 0, false>>::type;
  constexpr uint32_t QPW = g2_engine ? 8u : 16u;  // points per wave of the quad / octet kernels
  const bool quad_reduce_ok = pl.shared || pl.c <= 16;
#ifdef ZKMI_EXPERIMENTS
  const int accum_block = ZK_TUNE("ZKMI_ACCUM_BLOCK", 64) == 256 ? 256 : 64;
  const int accum_rounds = ZK_TUNE("ZKMI_ACCUM_ROUNDS", 0);
  auto striped = [&](uint32_t threads_needed, uint32_t block) {
    uint32_t blocks = (threads_needed + block - 1) / block;
    if (accum_rounds > 1) blocks = (blocks + accum_rounds - 1) / accum_rounds;
    return blocks ? blocks : 1u;
  };
#else
  if (!nocall) return hipErrorInvalidValue;  // (unreachable: the defaults above are compiled in)
#endif
  if (any_into && !nocall) return hipErrorInvalidValue;
  auto bslot_of = [&](int m) { return bucket_slots ? bucket_slots[m] : slots[m]; };
  auto into_of = [&](int m) { return !add_at_reduce && bslot_of(m) != slots[m]; };
  auto second_of = [&](int m) { return add_at_reduce && bslot_of(m) != slots[m]; };
  auto bk_of = [&](int m) { return buckets + (size_t)(into_of(m) ? bslot_of(m) : slots[m]) * cap_buckets; };
  // everything the reductions below would refuse is refused here, before the first launch: an early return between the
  // accumulation and k_accum_redo would leave a redo list behind that the slot's next MSM appends to
  {
    const uint32_t spw = pl.nb >> pl.seg_log;
    if ((uint64_t)pl.nwin * spw > seg_cap || (uint64_t)(2 + msm_seg_bits(pl)) * pl.nwin > SLOT_PTS || tot_b > cap_buckets)
      return hipErrorInvalidValue;
  }
  // [0] length, [1 ..] list, [cap_buckets + 1] ticket (k_accum_redo)
  auto redo_of = [&](int m) { return this->redo + (size_t)slots[m] * (cap_buckets + 2); };

  // ---- in front of the accumulation: heavy-bucket kernels of every MSM that runs them on another stream ----
  // On another stream they are queued BEFORE the accumulation: their workgroups (two waves of up to 330 registers, 28-56 KB
  // of LDS) are then placed while the SIMDs are still free; queued behind it they waited for the accumulation to drain.
  bool heavy_first[MSM_MULTI_MAX];
  hipStream_t heavy_stream[MSM_MULTI_MAX];
  // (A/B library, ZKMI_HEAVY_DEFER=1: one small proof / MSM by itself queues its heavy launches BEHIND the accumulation launch --
  // still ordered behind the sort only -- so that the accumulation, the longest link, is submitted ~50 us per MSM earlier.
  // Measured in round 6: no gain, 2^14 - 2^15 slightly worse; g2_split2_ab.txt)
  const bool heavy_deferred = host_spin && tot_b <= (1u << 16) && (quad_mask & 8) && heavy_on == 2 && ZK_TUNE("ZKMI_HEAVY_DEFER", 0) != 0;
  auto launch_heavy = [&](int m) {
    const MsmSort& sort = *sorts[m];
    // MSMs of different slots may overlap: partial sums, plan and tickets per slot
    XYZZ<F>* const hp = heavy_partial + (size_t)slots[m] * ((size_t)MSM_HPOOL + MSM_HNC_POOL);
    XYZZ<F>* const pool_nc = hp + MSM_HPOOL;
    uint32_t* const hplan = heavy_plan + (size_t)slots[m] * MSM_HPLAN_WORDS;
    uint32_t* const tk = heavy_ticket + (size_t)slots[m] * MSM_HEAVY_CAP;
    const bool wide_tail = !pl.shared && pl.c > 16;  // partitioned big windows: see k_accum_heavy
    // the call-free kernels take the additions of points (DESIGN.md 4.1 "heavy buckets"); not for lists the plan cannot
    // hold (decided on the device), the partitioned big windows' thousands of entries, or the A/B accumulate-into forms
    // (A/B library: ZKMI_HEAVY_NC=0 = the complete-law kernel of rounds 1-4 for everything)
    const bool nc = ZK_TUNE("ZKMI_HEAVY_NC", 1) != 0 && !wide_tail && !into_of(m);
    constexpr uint32_t lanes = std::is_same<F, Fq2_28>::value ? 32u : 64u;
    if (nc) {
      hipLaunchKernelGGL(k_heavy_plan<0>, dim3(1), dim3(64), 0, heavy_stream[m], sort.heavy, sort.count, sort.begin, hplan, lanes, 0u);
      // one-wave workgroups, grid-strided over the wave-items: twice the chip's wave slots (an empty list costs one placement)
      if constexpr (std::is_same<F, Fq2_28>::value)
        hipLaunchKernelGGL((k_accum_heavy_nc_g2<2>), dim3(4096), dim3(64), 0, heavy_stream[m], d_bases[m], sort.sorted, hplan, pool_nc);
      else
        hipLaunchKernelGGL((k_accum_heavy_nc<F, 3>), dim3(6144), dim3(64), 0, heavy_stream[m], d_bases[m], sort.sorted, hplan, pool_nc);
    }
    const uint32_t* const plan_arg = nc ? hplan : nullptr;
    if (nc && (quad_mask & 8)) {
      // both modes of the plan in one-wave workgroups of quads / octets: nothing of 300 registers has to find a SIMD beside the accumulation
      hipLaunchKernelGGL(k_heavy_q<QPT>, dim3(1024), dim3(64), 0, heavy_stream[m], d_bases[m], sort.begin, sort.count, sort.heavy,
                         sort.sorted, bk_of(m), hp, tk, hplan, pool_nc);
    } else if constexpr (std::is_same<F, Fq2_28>::value) {
      hipLaunchKernelGGL(k_accum_heavy_g2_split<0>, dim3(MSM_HSPLIT, 8), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T / 2, heavy_stream[m],
                         d_bases[m], sort.begin, sort.count, sort.heavy, sort.sorted, bk_of(m), hp, tk, 0u,
                         wide_tail ? MSM_HEAVY_CAP : 0xffffffffu, plan_arg, pool_nc);
      if (wide_tail)
        hipLaunchKernelGGL(k_accum_heavy_g2_split<0>, dim3(1, 4096), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T / 2, heavy_stream[m],
                           d_bases[m], sort.begin, sort.count, sort.heavy, sort.sorted, bk_of(m), hp, tk, MSM_HEAVY_CAP, 0xffffffffu,
                           (const uint32_t*)nullptr, pool_nc);
    } else {
      hipLaunchKernelGGL(k_accum_heavy<F>, dim3(MSM_HSPLIT, 8), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T, heavy_stream[m],
                         d_bases[m], sort.begin, sort.count, sort.heavy, sort.sorted, bk_of(m), hp, tk, into_of(m) ? 1u : 0u, 0u,
                         wide_tail ? MSM_HEAVY_CAP : 0xffffffffu, plan_arg, pool_nc);
      if (wide_tail)
        hipLaunchKernelGGL(k_accum_heavy<F>, dim3(1, 4096), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T, heavy_stream[m],
                           d_bases[m], sort.begin, sort.count, sort.heavy, sort.sorted, bk_of(m), hp, tk, into_of(m) ? 1u : 0u, MSM_HEAVY_CAP,
                           0xffffffffu, (const uint32_t*)nullptr, pool_nc);
    }
  };
  for (int m = 0; m < nm; m++) {
    slot_plan[slots[m]] = sorts[m]->plan;
    const hipStream_t st_reduce = st_reduces[m];
    const bool on_reduce = heavy_on == 2 && st_reduce != st;
    const bool side = !on_reduce && heavy_on != 0 && st_heavy && st_heavy != st;
    heavy_first[m] = side || on_reduce;
    heavy_stream[m] = on_reduce ? st_reduce : side ? st_heavy : st;
    if (into_of(m)) {
      // the bucket array is complete once the first MSM's redo pass has run (it follows that MSM's accumulation and
      // heavy-bucket kernels on its reduction stream)
      if ((e = hipStreamWaitEvent(st, redo_done[bslot_of(m)], 0)) != hipSuccess) return e;
      if (heavy_stream[m] != st && (e = hipStreamWaitEvent(heavy_stream[m], redo_done[bslot_of(m)], 0)) != hipSuccess) return e;
      if (st_reduce != st && st_reduce != heavy_stream[m] && (e = hipStreamWaitEvent(st_reduce, redo_done[bslot_of(m)], 0)) != hipSuccess) return e;
    }
    if (heavy_first[m]) {
      if ((e = hipEventRecord(pre[slots[m]], st)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(heavy_stream[m], pre[slots[m]], 0)) != hipSuccess) return e;
      if (!heavy_deferred) launch_heavy(m);
    }
  }
  auto view_of = [&](int m) {
    const MsmSort& sort = *sorts[m];
    return SortView{sort.begin, sort.count, sort.perm, sort.sorted, sort.plan.heavy_thr};
  };

  // ---- the accumulation ----
  // A/B library test switch ZKMI_FORCE_MULTI: a lone MSM through the multi-MSM launch too (so that the generic MSM entry
  // points, with their degenerate inputs -- repeated bases, P + (-P), points at infinity --, reach k_accum_g1_split2 / the
  // MULTI kernel form)
  const bool force_multi = ZK_TUNE("ZKMI_FORCE_MULTI", 0) != 0;
  if (prof) prof->begin(ph_accum, st);
  if constexpr (std::is_same<F, Fq2_28>::value) {
    for (int m = 0; m < nm; m++) {
      const MsmSort& sort = *sorts[m];
      const uint32_t heavy_thr = sort.plan.heavy_thr;
      XYZZ<F>* const bk = bk_of(m);
      uint32_t* const redo = redo_of(m);
#ifdef ZKMI_EXPERIMENTS
      if (mode == 3) {
        hipLaunchKernelGGL((k_accum_g2_nc<3, 1>), dim3((2 * tot_b + 63) / 64), dim3(64), 0, st, d_bases[m], sort.begin, sort.count,
                           sort.perm, sort.sorted, bk, tot_b, heavy_thr, redo);
        continue;
      }
      if (mode != 2) {
        if (accum_block == 64)
          hipLaunchKernelGGL(k_accum_g2_split<1>, dim3(striped(2 * tot_b, 64)), dim3(64), 0, st, d_bases[m], sort.begin, sort.count,
                             sort.perm, sort.sorted, bk, tot_b, heavy_thr);
        else
          hipLaunchKernelGGL(k_accum_g2_split<4>, dim3(striped(2 * tot_b, 256)), dim3(256), 0, st, d_bases[m], sort.begin, sort.count,
                             sort.perm, sort.sorted, bk, tot_b, heavy_thr);
        continue;
      }
#endif
#ifdef ZKMI_EXPERIMENTS
      if ((quad_mask & 16) && tot_b <= (1u << 16)) {
        // one small MSM by itself: an octet per bucket (msm_quad.hpp k_accum_q) -- 4 products per entry instead of 10
        if constexpr (g2_engine) {
          AccumQArgs<QPT> qa;
          for (int k = 0; k < MSM_MULTI_MAX; k++) {
            qa.bases[k] = d_bases[m];
            qa.buckets[k] = bk;
            qa.sort[k] = view_of(m);
          }
          hipLaunchKernelGGL(k_accum_q<QPT>, dim3((tot_b + QPW - 1) / QPW, 1), dim3(64), 0, st, qa, tot_b);
        }
        continue;
      }
#endif
      if (host_spin && tot_b <= (1u << 15) && ZK_TUNE("ZKMI_SOLO_SPLIT_G2", 1) != 0) {
        // one small G2 MSM by itself (the G2 launch of a single small proof): two lane pairs per bucket halve the chain
        // (2^13 1.99 -> 1.74 ms, 2^14 1.92 -> 1.77, 2^15 2.38 -> 2.2; at 2^16 buckets the chip is full and it costs 0.08 ms:
        // profiles/r06/experiments/g2_split2_ab.txt)
        hipLaunchKernelGGL(k_accum_g2_split2<0>, dim3((4 * tot_b + 63) / 64), dim3(64), 0, st, d_bases[m], sort.begin, sort.count,
                           sort.perm, sort.sorted, bk, tot_b, heavy_thr, redo);
        continue;
      }
      hipLaunchKernelGGL((k_accum_g2_nc<2, 1>), dim3((2 * tot_b + 63) / 64), dim3(64), 0, st, d_bases[m], sort.begin, sort.count,
                         sort.perm, sort.sorted, bk, tot_b, heavy_thr, redo);
    }
  } else if ((nm > 1 || (force_multi && !any_into)) && nocall) {
    AccumArgs<F, true> set;
    for (int m = 0; m < MSM_MULTI_MAX; m++) {
      const int k = m < nm ? m : 0;
      set.bases_[m] = d_bases[k];
      set.buckets_[m] = bk_of(k);
      set.redo_[m] = redo_of(k);
      set.sort_[m] = view_of(k);
    }
    // small plans (one small proof): two lanes per bucket (A/B library: ZKMI_SOLO_SPLIT=0: one)
    const bool split2 = ZK_TUNE("ZKMI_SOLO_SPLIT", 1) != 0;
#ifdef ZKMI_EXPERIMENTS
    if ((quad_mask & 16) && tot_b <= (1u << 16)) {
      // a quad per bucket (msm_quad.hpp k_accum_q): the complete law in line, no redo list
      if constexpr (!g2_engine) {
        AccumQArgs<QPT> qa;
        for (int m = 0; m < MSM_MULTI_MAX; m++) {
          const int k = m < nm ? m : 0;
          qa.bases[m] = d_bases[k];
          qa.buckets[m] = bk_of(k);
          qa.sort[m] = view_of(k);
        }
        hipLaunchKernelGGL(k_accum_q<QPT>, dim3((tot_b + QPW - 1) / QPW, nm), dim3(64), 0, st, qa, tot_b);
      }
    } else
#endif
    if (split2 && tot_b <= (1u << 16))
      hipLaunchKernelGGL((k_accum_g1_split2<F, 1>), dim3((2 * tot_b + 63) / 64, nm), dim3(64), 0, st, set, tot_b);
#ifdef ZKMI_EXPERIMENTS
    else if (accum_mode != 3)
      hipLaunchKernelGGL((k_accum_g1_nc<F, 2, 1, true>), dim3((tot_b + 63) / 64, nm), dim3(64), 0, st, set, tot_b);
#endif
    else
      hipLaunchKernelGGL((k_accum_g1_nc<F, 3, 1, true>), dim3((tot_b + 63) / 64, nm), dim3(64), 0, st, set, tot_b);
  } else {
    for (int m = 0; m < nm; m++) {
      const MsmSort& sort = *sorts[m];
      XYZZ<F>* const bk = bk_of(m);
      uint32_t* const redo = redo_of(m);
      const AccumArgs<F, false> one = {d_bases[m], bk, redo, view_of(m)};
      if (into_of(m)) {
#ifdef ZKMI_EXPERIMENTS
        hipLaunchKernelGGL((k_accum_g1_nc<F, 3, 1, false, true>), dim3((tot_b + 63) / 64), dim3(64), 0, st, one, tot_b);
        continue;
#else
        return hipErrorInvalidValue;  // (the accumulate-into kernel is an A/B variant: the product adds at the reduction)
#endif
      }
#ifdef ZKMI_EXPERIMENTS
      if (accum_mode != 3) {
        const uint32_t heavy_thr = sort.plan.heavy_thr;
        const dim3 grid((tot_b + T - 1) / T);
        if (accum_mode == 2)
          hipLaunchKernelGGL((k_accum_g1_nc<F, 2, 1>), dim3((tot_b + 63) / 64), dim3(64), 0, st, one, tot_b);
        else if (accum_mode == 1)
          hipLaunchKernelGGL(k_accum<F>, grid, dim3(T), 0, st, d_bases[m], sort.begin, sort.count, sort.perm, sort.sorted, bk, tot_b,
                             heavy_thr);
        else if (accum_block == 64)
          hipLaunchKernelGGL((k_accum_g1_glds<F, 1>), dim3(striped(tot_b, 64)), dim3(64), 0, st, d_bases[m], sort.begin, sort.count,
                             sort.perm, sort.sorted, bk, tot_b, heavy_thr);
        else
          hipLaunchKernelGGL((k_accum_g1_glds<F, 4>), dim3(striped(tot_b, 256)), dim3(256), 0, st, d_bases[m], sort.begin, sort.count,
                             sort.perm, sort.sorted, bk, tot_b, heavy_thr);
        continue;
      }
#endif
      (void)sort;
#ifdef ZKMI_EXPERIMENTS
      // (A/B library: the occupancy-capped kernel of the pipelined big MSM -- msm_pipe.hpp, measured neutral -- or, with
      // ZKMI_ACCUM_W2=1, for every lone G1 accumulation: 12 % slower than three waves)
      if ((flags & MSM_RUN_TWO_WAVES) || ZK_TUNE("ZKMI_ACCUM_W2", 0) == 1) {
        hipLaunchKernelGGL(k_accum_g1_nc_w2<F>, dim3((tot_b + 63) / 64), dim3(64), 0, st, one, tot_b);
        continue;
      }
#endif
      hipLaunchKernelGGL((k_accum_g1_nc<F, 3, 1>), dim3((tot_b + 63) / 64), dim3(64), 0, st, one, tot_b);
    }
  }
  if (prof) prof->end(ph_accum, st);  // the phase brackets the accumulation launch(es) only (roofline leg of bench.py)
  if (heavy_deferred)
    for (int m = 0; m < nm; m++)
      if (heavy_first[m]) launch_heavy(m);

  // ---- behind it, per MSM: heavy kernels that run in line, redo list, reduction, copy of the partials ----
  const uint32_t segs_per_win = pl.nb >> pl.seg_log;
  const uint32_t tot_segs = pl.nwin * segs_per_win;
  const int seg = 1 << pl.seg_log;
  const int plain_job = pl.shared ? 1 + msm_seg_bits(pl) : -1;
  const int njobs = 1 + msm_seg_bits(pl) + (pl.shared ? 1 : 0);
  // small plans: the job lists are cut into slices of two segments per thread or lane pair (k_treesum_final adds the slices)
  constexpr bool is_g2 = std::is_same<F, Fq2_28>::value;
  const uint32_t per_block = is_g2 ? MSM_TREE_T : 2 * MSM_TREE_T;  // G2: 64 lane pairs x 2 segments
  uint32_t nchunk = 1;
  if (tot_b <= (1u << 16) && segs_per_win > per_block) {
    nchunk = segs_per_win / per_block;
    if (nchunk > MSM_TREE_T) nchunk = MSM_TREE_T;
    // (fewer, longer slices -- at most 16, 8 or 4 -- measured the same group rates and single-proof latencies within noise:
    // profiles/r05/experiments/tree_slices_ab.txt)
    if ((uint64_t)njobs * pl.nwin * nchunk > MSM_STAGE_PTS_1) nchunk = 1;
  } else if (!pl.shared && tot_b <= (1u << 19) && segs_per_win > per_block) {
    // one windowed MSM of up to 2^19 buckets: as many slices as the stage holds (16 windows x 13 jobs: 16), see plan_set_heavy
    // (... and as the chip holds at once: 3 328 workgroups of 9 dependent additions each, in three rounds, took longer than 208 of 23)
    nchunk = MSM_TREE_T;
    while (nchunk > 1 && ((uint64_t)njobs * pl.nwin * nchunk > 1024 || nchunk * per_block > segs_per_win)) nchunk >>= 1;
  } else if (!pl.shared && pl.c > 16) {
    // partitioned big windows: 2^(c-5) segments per (window, job) list on ONE workgroup are a chain of 2^(c-12) dependent
    // additions (5 ms at c = 20); sliced, as many slices as the stage holds (16 at 13 windows x 16 jobs)
    nchunk = MSM_TREE_T;
    while (nchunk > 1 && ((uint64_t)njobs * pl.nwin * nchunk > MSM_STAGE_PTS_1 || nchunk * per_block > segs_per_win)) nchunk >>= 1;
  }
  for (int m = 0; m < nm; m++) {
    const MsmSort& sort = *sorts[m];
    const int slot = slots[m];
    const hipStream_t st_reduce = st_reduces[m];
    XYZZ<F>* const bk = bk_of(m);
    if (!heavy_first[m]) launch_heavy(m);
    if ((e = hipEventRecord(acc_done[slot], st)) != hipSuccess) return e;
    sort.readers.push_back(acc_done[slot]);  // the next sort into these buffers may be queued on another stream
    if (st_reduce != st && (e = hipStreamWaitEvent(st_reduce, acc_done[slot], 0)) != hipSuccess) return e;
    if (heavy_first[m] && heavy_stream[m] != st_reduce) {  // side stream
      if ((e = hipEventRecord(heavy_done[slot], heavy_stream[m])) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(st_reduce, heavy_done[slot], 0)) != hipSuccess) return e;
      sort.readers.push_back(heavy_done[slot]);  // the next sort must not overwrite what these kernels read
    }
    if (heavy_first[m] && heavy_stream[m] == st_reduce && !nocall) {  // (with the call-free kernels redo_done below covers the heavy kernels too: same stream, later)
      if ((e = hipEventRecord(heavy_done[slot], st_reduce)) != hipSuccess) return e;
      sort.readers.push_back(heavy_done[slot]);
    }
    if (nocall) {
      // after the accumulation (it writes the list), in front of the reduction (it reads the buckets);
      // heavy buckets are never listed, so the heavy kernels may still be running
      uint32_t* const redo = redo_of(m);
      if (quad_mask & 4)
        hipLaunchKernelGGL(k_accum_redo_q<QPT>, dim3(64), dim3(64), 0, st_reduce, d_bases[m], sort.begin, sort.count, sort.sorted, bk, redo,
                           redo + cap_buckets + 1, into_of(m) ? 1u : 0u);
      else if constexpr (std::is_same<F, Fq2_28>::value)
        hipLaunchKernelGGL(k_accum_redo_g2_split<0>, dim3(64), dim3(64), sizeof(XYZZ<F>) * 32, st_reduce, d_bases[m], sort.begin, sort.count, sort.sorted, bk,
                           redo, redo + cap_buckets + 1);
      else
        hipLaunchKernelGGL(k_accum_redo<F>, dim3(64), dim3(64), sizeof(XYZZ<F>) * 64, st_reduce, d_bases[m], sort.begin, sort.count, sort.sorted, bk, redo,
                           redo + cap_buckets + 1, into_of(m) ? 1u : 0u);
      // the list reads the sort: the next sort must wait for this kernel too
      if ((e = hipEventRecord(redo_done[slot], st_reduce)) != hipSuccess) return e;
      sort.readers.push_back(redo_done[slot]);
    }
    if (no_reduce) continue;  // a later MSM adds into these buckets and reduces them (bucket_slots)
    if (prof) prof->begin(ph_reduce, st_reduce);
    XYZZ<F>* const ssum = segsum + (size_t)slot * seg_cap;
    XYZZ<F>* const sw = segw + (size_t)slot * seg_cap;
    if constexpr (is_g2) {
      if ((quad_mask & 1) && quad_reduce_ok)
        hipLaunchKernelGGL(k_segreduce_q<QPT>, dim3((tot_segs + QPW - 1) / QPW), dim3(64), 0, st_reduce, bk, ssum, sw, tot_segs, seg,
                           (const XYZZ<F>*)nullptr, (const XYZZ<F>*)nullptr);
      else
        hipLaunchKernelGGL(k_segreduce_g2_split<0>, dim3((2 * tot_segs + T - 1) / T), dim3(T), 0, st_reduce, bk, ssum, sw,
                           tot_segs, seg);
    } else {
      const XYZZ<F>*bk2 = nullptr, *bk3 = nullptr;
      if (second_of(m)) {
        // the other MSM's bucket array is complete behind its redo pass (which follows its accumulation and heavy-bucket kernels)
        if ((e = hipStreamWaitEvent(st_reduce, redo_done[bslot_of(m)], 0)) != hipSuccess) return e;
        bk2 = buckets + (size_t)bslot_of(m) * cap_buckets;
        if (third_slot >= 0) {
          if ((e = hipStreamWaitEvent(st_reduce, redo_done[third_slot], 0)) != hipSuccess) return e;
          bk3 = buckets + (size_t)third_slot * cap_buckets;
        }
      }
      if (!pl.shared && pl.c > 16)
        hipLaunchKernelGGL(k_segreduce_w2<F>, dim3((tot_segs + T - 1) / T), dim3(T), 0, st_reduce, bk, ssum, sw, tot_segs, seg, bk2, bk3);
      else if (quad_mask & 1)
        hipLaunchKernelGGL(k_segreduce_q<QPT>, dim3((tot_segs + QPW - 1) / QPW), dim3(64), 0, st_reduce, bk, ssum, sw, tot_segs, seg, bk2, bk3);
      else
        hipLaunchKernelGGL(k_segreduce<F>, dim3((tot_segs + T - 1) / T), dim3(T), 0, st_reduce, bk, ssum, sw, tot_segs, seg, bk2, bk3);
    }
    XYZZ<HF>* dp = partial + (size_t)slot * SLOT_PTS;
    XYZZ<HF>* const hp_out = h_partial + (size_t)slot * SLOT_PTS;  // pinned host slot, written by the tree-sum kernels themselves
    XYZZ<F>* const stg = tree_stage + (size_t)slot * MSM_STAGE_PTS;
    if ((quad_mask & 2) && quad_reduce_ok) {
      // one wave (16 quads / 8 octets) per slice of a job list; as many slices as make the two levels equally deep (a point
      // walks len / (PER_WAVE slices) segments, the final sum slices / PER_WAVE), within the stage
      uint32_t nq = 1;
      while ((uint64_t)nq * nq * 4 <= segs_per_win) nq <<= 1;  // ~ sqrt(segs_per_win), rounded to a power of two
      while (nq > 1 && ((uint64_t)njobs * pl.nwin * nq > MSM_STAGE_PTS || nq * QPW > segs_per_win)) nq >>= 1;
      hipLaunchKernelGGL(k_treesum_q<QPT>, dim3(njobs, pl.nwin, nq), dim3(64), 0, st_reduce, ssum, sw, segs_per_win, dp, plain_job, stg, hp_out);
      if (nq > 1) hipLaunchKernelGGL(k_treesum_final_q<QPT>, dim3(njobs, pl.nwin), dim3(64), 0, st_reduce, stg, nq, dp, hp_out);
    } else if constexpr (is_g2) {
      if (nchunk > 1) {
        hipLaunchKernelGGL(k_treesum_g2_split<0>, dim3(njobs, pl.nwin, nchunk), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T / 2, st_reduce,
                           ssum, sw, segs_per_win, dp, plain_job, stg, hp_out);
        uint32_t tf = 64;
        while (tf < 2 * nchunk) tf <<= 1;
        hipLaunchKernelGGL(k_treesum_final_g2_split<0>, dim3(njobs, pl.nwin), dim3(tf), sizeof(XYZZ<F>) * tf / 2, st_reduce, stg, nchunk, dp, hp_out);
      } else {
        // big plans: the same lane-pair kernel, 128 pairs per job (the unsplit form's 330-register additions made this the
        // slowest link of a proof's tail: 16 + 7 dependent additions of 45-60 us); A/B library, ZKMI_G2_TREE_SPLIT=0: the unsplit kernel
#ifdef ZKMI_EXPERIMENTS
        if (ZK_TUNE("ZKMI_G2_TREE_SPLIT", 1) == 0)
          hipLaunchKernelGGL(k_treesum<F>, dim3(njobs, pl.nwin, 1), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T, st_reduce,
                             ssum, sw, segs_per_win, dp, plain_job, stg, hp_out);
        else
#endif
          hipLaunchKernelGGL(k_treesum_g2_split<0>, dim3(njobs, pl.nwin, 1), dim3(2 * MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T, st_reduce,
                             ssum, sw, segs_per_win, dp, plain_job, stg, hp_out);
      }
    } else {
      hipLaunchKernelGGL(k_treesum<F>, dim3(njobs, pl.nwin, nchunk), dim3(MSM_TREE_T), sizeof(XYZZ<F>) * MSM_TREE_T, st_reduce,
                         ssum, sw, segs_per_win, dp, plain_job, stg, hp_out);
      if (nchunk > 1) {
        uint32_t tf = 64;
        while (tf < nchunk) tf <<= 1;
        hipLaunchKernelGGL(k_treesum_final<F>, dim3(njobs, pl.nwin), dim3(tf), sizeof(XYZZ<F>) * tf, st_reduce, stg, nchunk, dp, hp_out);
      }
    }
    if (prof) prof->end(ph_reduce, st_reduce);
    // (the partials are in the pinned host slot when the last tree-sum kernel has finished: the event is all that is left)
    if ((e = hipEventRecord(done[slot], st_reduce)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

template <class F>
hipError_t MsmEngine<F>::finish_host_windows(XYZZ<HF>* out_windows, int slot) {
  hipError_t e = wait_event(done[slot], host_spin);  // (host_pool.hpp: the batch prover's driving thread polls, then sleeps)
  if (e != hipSuccess) return e;
  // (host_spin = one MSM or one proof by itself: the latency path)
  windows_from_partials(slot_plan[slot], h_partial + (size_t)slot * SLOT_PTS, out_windows, host_spin);
  return hipSuccess;
}

// the per-(window, job) sums a reduction leaves behind -> one sum per window (host arithmetic).  `h` may come from this
// device's pinned slot or from another rank's copy of the same array (comm.hip: the all-gathered partials of a point split).
template <class F>
int MsmEngine<F>::partials_per_msm(const MsmPlan& pl) { return pl.nwin * (1 + msm_seg_bits(pl) + (pl.shared ? 1 : 0)); }
template <class F>
void MsmEngine<F>::windows_from_partials(const MsmPlan& pl, const XYZZ<HF>* h, XYZZ<HF>* out_windows, bool parallel) {
  const int seg_bits = msm_seg_bits(pl);
  const int njobs = 1 + seg_bits + (pl.shared ? 1 : 0);
  const std::function<void(uint32_t)> one = [&](uint32_t w) {
    XYZZ<HF> u = XYZZ<HF>::infinity();
    // (top window of a partitioned big-window plan: the top bits of the segment index number the partition its entries
    // were spread to, not the digit -- MsmPlan::top_spread_log)
    const int bits = (pl.win_first + (int)w == pl.total_windows() - 1 && pl.top_spread_log > 0) ? seg_bits - pl.top_spread_log : seg_bits;
    for (int j = bits - 1; j >= 0; j--) {
      u.dbl_inplace();
      u.add(h[(size_t)w * njobs + 1 + j]);
    }
    for (int i = 0; i < pl.seg_log; i++) u.dbl_inplace();
    u.add(h[(size_t)w * njobs]);
    out_windows[w] = u;
  };
  // One MSM by itself: its 13-16 windows (or partitions) are ~30 group operations each -- 0.3 ms on one thread between the
  // last kernel and the result, a seventh of a witness-like 2^20-term MSM; on the process's pool they run side by side.
  if (parallel && pl.nwin >= 8) {
    HostPool::instance().run((uint32_t)pl.nwin, host_cpu_budget(), one);
  } else {
    for (int w = 0; w < pl.nwin; w++) one((uint32_t)w);
  }
}

// shared buckets: partition q of a vector holds buckets q*nb + j (digit value q*nb + j + 1):
//   sum_b (b+1) B_b = sum_q U_q + nb * sum_q q * S_q,   U_q = weighted sum, S_q = plain sum of partition q
template <class HF>
static XYZZ<HF> msm_combine_partitions(const XYZZ<HF>* win, const XYZZ<HF>* h, int parts, int njobs, uint32_t nb) {
  XYZZ<HF> run = XYZZ<HF>::infinity(), t = XYZZ<HF>::infinity(), total = XYZZ<HF>::infinity();
  for (int q = parts - 1; q >= 1; q--) {
    run.add(h[(size_t)q * njobs + (njobs - 1)]);
    t.add(run);
  }
  for (uint32_t b = nb; b > 1; b >>= 1) t.dbl_inplace();
  for (int q = 0; q < parts; q++) total.add(win[q]);
  total.add(t);
  return total;
}

template <class F>
hipError_t MsmEngine<F>::finish_host_batch(XYZZ<HF>* out, int slot) {
  // the batched plan holds, per scalar vector, the vec_parts partitions of that vector's own bucket set
  const MsmPlan& pl = slot_plan[slot];
  if (pl.vec_parts <= 1) return finish_host_windows(out, slot);
  std::vector<XYZZ<HF>> win(pl.nwin);
  hipError_t e = finish_host_windows(win.data(), slot);
  if (e != hipSuccess) return e;
  const int njobs = 2 + msm_seg_bits(pl);
  const XYZZ<HF>* h = h_partial + (size_t)slot * SLOT_PTS;
  for (int v = 0; v * pl.vec_parts < pl.nwin; v++)
    out[v] = msm_combine_partitions<HF>(win.data() + (size_t)v * pl.vec_parts, h + (size_t)v * pl.vec_parts * njobs, pl.vec_parts,
                                        njobs, pl.nb);
  return hipSuccess;
}

template <class F>
hipError_t MsmEngine<F>::wait_slot(int slot) {
  return wait_event(done[slot], host_spin);
}
template <class F>
XYZZ<typename MsmEngine<F>::HF> MsmEngine<F>::host_result_vec(int slot, int v) const {
  const MsmPlan& pl = slot_plan[slot];
  const int vp = pl.vec_parts > 1 ? pl.vec_parts : 1;
  const int njobs = 1 + msm_seg_bits(pl) + (pl.shared ? 1 : 0);
  const XYZZ<HF>* h = h_partial + (size_t)slot * SLOT_PTS + (size_t)v * vp * njobs;
  MsmPlan mine = pl;  // the partitions of this vector only
  mine.nwin = vp;
  XYZZ<HF> win[64];
  windows_from_partials(mine, h, win);
  if (vp == 1) return win[0];
  return msm_combine_partitions<HF>(win, h, vp, njobs, pl.nb);
}

template <class HF>
XYZZ<HF> msm_combine_windows(const XYZZ<HF>* windows, int nwin, int c) {
  XYZZ<HF> total = XYZZ<HF>::infinity();
  for (int w = nwin - 1; w >= 0; w--) {
    for (int i = 0; i < c; i++) total.dbl_inplace();
    total.add(windows[w]);
  }
  return total;
}

template <class F>
hipError_t MsmEngine<F>::finish_host(XYZZ<HF>* out, int slot) {
  const MsmPlan& pl = slot_plan[slot];
  std::vector<XYZZ<HF>> win(pl.nwin);
  hipError_t e = finish_host_windows(win.data(), slot);
  if (e != hipSuccess) return e;
  if (!pl.shared) {
    *out = msm_combine_windows(win.data(), pl.nwin, pl.c);
    return hipSuccess;
  }
  *out = msm_combine_partitions<HF>(win.data(), h_partial + (size_t)slot * SLOT_PTS, pl.nwin, 2 + msm_seg_bits(pl), pl.nb);
  return hipSuccess;
}

// table[w * n + i] = 2^(c w) * bases[i]: one thread per base point walks the digits
template <class F>
__global__ void __launch_bounds__(64)
k_build_table(const Affine<F>* __restrict__ bases, Affine<F>* __restrict__ table, uint32_t n, int ndigits, int c) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> p = load_vec(bases + i);
  XYZZ<F> cur = XYZZ<F>::from_affine(p);
  store_vec(table + i, p);
  for (int w = 1; w < ndigits; w++) {
    for (int k = 0; k < c; k++) cur.dbl_inplace();
    const Affine<F> a = cur.to_affine();
    // points at infinity must be exact zeros (the accumulate kernels test limbs)
    Affine<F> o = cur.is_inf() ? Affine<F>::infinity() : a;
    store_vec(table + (size_t)w * n + i, o);
  }
}

template <class F>
hipError_t msm_build_table(const Affine<F>* d_bases, uint64_t n, const MsmPlan& plan, Affine<F>** out_table,
                           hipStream_t st) {
  *out_table = nullptr;
  hipError_t e = hipMalloc(out_table, sizeof(Affine<F>) * (size_t)plan.ndigits * (n ? n : 1));
  if (e != hipSuccess) return e;
  if (n) {
    hipLaunchKernelGGL(k_build_table<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_bases, *out_table,
                       (uint32_t)n, plan.ndigits, plan.c);
    e = hipGetLastError();
  }
  return e;
}

template <class F>
hipError_t bases_convert(const Affine<typename HostFieldOf<F>::type>* d_in, Affine<F>* d_out, uint64_t n,
                         hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_bases_convert<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, d_out, (uint32_t)n);
  return hipGetLastError();
}

}  // namespace zkmi

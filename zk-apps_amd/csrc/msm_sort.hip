// zkmi — bucket scatter for the Pippenger MSM: signed-digit decomposition of
// the scalars and a counting sort of point indices by (window, bucket).
// See msm_impl.hpp for the full kernel chain and HBM layout.
#include <stdio.h>
#include <stdlib.h>
#include "msm_impl.hpp"
#include "tune.hpp"

namespace zkmi {

namespace {

// Signed-digit recoding without a carry chain: with
//   M = sum_w (2^(c-1) - 1) 2^(c w)
// the unsigned base-2^c digits e_w of k' = k + M give d_w = e_w - (2^(c-1) - 1)
// in [-(2^(c-1) - 1), 2^(c-1)], so any (scalar, window) digit is independent.
struct RecodeConst {
  uint32_t m[9];
};

__device__ __forceinline__ void load_biased(const uint32_t* __restrict__ scalars, uint32_t i, const RecodeConst& rc,
                                            uint32_t* k) {
  const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = q[0], b = q[1];
  const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint64_t v = (uint64_t)s[j] + rc.m[j] + carry;
    k[j] = (uint32_t)v;
    carry = v >> 32;
  }
  k[8] = rc.m[8] + (uint32_t)carry;
}

// returns bucket index + 1 (0 = skip) and the sign
__device__ __forceinline__ uint32_t digit_of(const uint32_t* k, int w, int c, bool& neg) {
  const int bit = w * c;
  const int limb = bit >> 5, sh = bit & 31;
  uint64_t v = k[limb];
  if (limb + 1 < 9) v |= (uint64_t)k[limb + 1] << 32;
  const int32_t e = (int32_t)((uint32_t)(v >> sh) & ((1u << c) - 1u));
  const int32_t d = e - (int32_t)((1u << (c - 1)) - 1u);
  neg = d < 0;
  return (uint32_t)(neg ? -d : d);
}

// Pass 1 (SCATTER = false): LDS histogram of one (window, chunk) tile -> blockhist.
// Pass 3 (SCATTER = true):  blockhist holds the tile's first output slot per
// bucket; LDS cursors hand out slots (ds_add_rtn), point ids go to sorted[].
template <bool SCATTER>
__global__ void __launch_bounds__(1024)
k_bucket_pass(const uint32_t* __restrict__ scalars, uint32_t n, int c, uint32_t nb, uint32_t chunk, RecodeConst rc,
              uint32_t* __restrict__ blockhist, uint32_t* __restrict__ sorted, uint32_t w0) {
  extern __shared__ uint32_t hist[];
  const uint32_t ch = blockIdx.x, w = blockIdx.y, nch = gridDim.x;  // w: window of this sort; w0 + w: digit position
  uint32_t* gh = blockhist + ((size_t)w * nch + ch) * nb;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = SCATTER ? gh[b] : 0u;
  __syncthreads();
  const uint32_t beg = ch * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    uint32_t k[9];
    load_biased(scalars, i, rc, k);
    bool neg;
    const uint32_t d = digit_of(k, (int)(w0 + w), c, neg);
    if (d) {
      if (SCATTER) {
        const uint32_t pos = atomicAdd(&hist[d - 1], 1u);
        sorted[pos] = i | (neg ? 0x80000000u : 0u);
      } else {
        atomicAdd(&hist[d - 1], 1u);
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) gh[b] = hist[b];
  }
}

// Shared-bucket mode: one bucket set of P * nb buckets for all digits.  Tile (chunk, q)
// scans its scalars' ndigits digits and keeps those whose bucket falls in partition q.
template <bool SCATTER>
__global__ void __launch_bounds__(1024)
k_bucket_pass_shared(const uint32_t* __restrict__ scalars, uint32_t n, int c, int ndigits, uint32_t nb, int nb_log,
                     uint32_t chunk, RecodeConst rc, uint32_t* __restrict__ blockhist, uint32_t* __restrict__ sorted,
                     uint64_t batch_stride_words, uint32_t vec_parts) {
  extern __shared__ uint32_t hist[];
  const uint32_t ch = blockIdx.x, q = blockIdx.y, nch = gridDim.x;
  // batch mode (batch_stride_words != 0): partition q of the combined bucket array is bucket range q % vec_parts of
  // the independent scalar vector q / vec_parts; table indices are per vector
  const uint32_t vec = q / vec_parts, range = q - vec * vec_parts;
  scalars += (size_t)vec * batch_stride_words;
  uint32_t* gh = blockhist + ((size_t)q * nch + ch) * nb;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = SCATTER ? gh[b] : 0u;
  __syncthreads();
  const uint32_t beg = ch * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    uint32_t k[9];
    load_biased(scalars, i, rc, k);
    for (int w = 0; w < ndigits; w++) {
      bool neg;
      const uint32_t d = digit_of(k, w, c, neg);
      if (d == 0) continue;
      const uint32_t bkt = d - 1;
      if ((bkt >> nb_log) != range) continue;
      const uint32_t local = bkt & (nb - 1u);
      if (SCATTER) {
        const uint32_t pos = atomicAdd(&hist[local], 1u);
        sorted[pos] = ((uint32_t)w * n + i) | (neg ? 0x80000000u : 0u);
      } else {
        atomicAdd(&hist[local], 1u);
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) gh[b] = hist[b];
  }
}

// LDS counter updates of one wave, aggregated when all its active lanes name the SAME counter: a witness of bits sends
// 630 000 records to one bucket, i.e. 64-way same-address ds_add conflicts on every wave-instruction of both passes (with
// the round structure below: 3.7 ms of k_fpart_sort for a 2^20 bit witness, 0.15 ms for a dense one).  Returns the value
// the lane's own atomicAdd(&ctr[idx], 1) would have returned (any order within the wave is as good as another).
__device__ __forceinline__ uint32_t wave_counter_add(uint32_t* ctr, uint32_t idx, bool valid) {
  const uint64_t mask = __ballot(valid);
  if (mask == 0) return 0u;
  const int leader = __ffsll((unsigned long long)mask) - 1;
  const uint32_t idx0 = (uint32_t)__shfl((int)idx, leader);
  if (__ballot(valid && idx != idx0) == 0) {  // wave-uniform
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&ctr[idx0], (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << (threadIdx.x & 63u)) - 1ull));
  }
  return valid ? atomicAdd(&ctr[idx], 1u) : 0u;
}

// ---- record pre-pass (shared mode with several partitions) -----------------------
// Scanning every scalar's digits once per partition costs P x the digit extraction.  With
// P > 1 the digits are extracted ONCE into (table index, local bucket) records grouped by
// partition; the LDS-histogram passes then stream their own partition's records.
// WIN = false: shared-bucket plan, P partitions of ONE bucket set, entries are table indices (digit * n + point).
// WIN = true: windowed plan with more than 2^15 buckets per window (c > 16: big inputs, where fewer, wider digits pay
// for their bigger bucket sets): digit w owns its own bucket set, cut into ppw = 2^(c-1-nb_log) partitions -- partition
// w * ppw + (bucket >> nb_log) of P = nwin * ppw -- and entries are plain point indices.  Either way every scalar is
// read ONCE per pass for all its digits (the plain windowed sort reads it once per window and pass).
constexpr uint32_t PART_MAX = 256;  // record groups of one sort: partitions (shared) / windows x partitions (windowed)
constexpr int FINE_LOG = 10;        // buckets per fine partition (the fine-partition sorts below)
// FINE (counting pass of the big windowed plans): also the records per FINE partition -- 2^fine_log consecutive buckets of a
// group -- summed over all blocks into fine_tot[group * (nb >> fine_log) + fine] (zeroed by the caller; dynamic LDS: one
// counter per fine partition)
// K: type of the stored bucket ids (uint16_t in the two-level sort: a record is 6 bytes instead of 8 through three passes)
template <bool WRITE, bool WIN = false, bool FINE = false, class K = uint32_t>
__global__ void __launch_bounds__(1024)
k_part_pass(const uint32_t* __restrict__ scalars, uint32_t n, int c, int ndigits, uint32_t nb, int nb_log, uint32_t P,
            uint32_t chunk, RecodeConst rc, uint32_t* __restrict__ blkcnt, uint32_t* __restrict__ rec_entry,
            K* __restrict__ rec_bkt, int w0 = 0, int w_top_pos = -1, uint32_t* __restrict__ fine_tot = nullptr,
            int fine_log = FINE_LOG) {
  __shared__ uint32_t cnt[PART_MAX];
  extern __shared__ uint32_t fine_cnt[];
  const int fan_log = nb_log - fine_log;
  if (threadIdx.x < PART_MAX) cnt[threadIdx.x] = (WRITE && threadIdx.x < P) ? blkcnt[blockIdx.x * P + threadIdx.x] : 0u;
  if (FINE)
    for (uint32_t f = threadIdx.x; f < (P << fan_log); f += blockDim.x) fine_cnt[f] = 0u;
  __syncthreads();
  const uint32_t ppw = WIN ? ((1u << (c - 1)) >> nb_log) : 0u;
  // the top window's digits are < 2^nb_log: its entries go to partition (point mod ppw) instead of partition 0 (MsmPlan::top_spread_log)
  // (WIN: digit positions [w0, w0 + ndigits) of the scalar; w_top_pos = position of the plan's top window)
  const int w_top = WIN ? w_top_pos - w0 : -1;
  const uint32_t beg = blockIdx.x * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  if (WIN && ppw == 1) {
    // one group per window (c = 16): at digit w every lane of a wave names counter w -- one LDS atomic per wave instead of a
    // 64-way same-address conflict (whole waves walk the scalars: the aggregated update needs every lane)
    for (uint32_t i0 = beg; i0 < end; i0 += blockDim.x) {
      const uint32_t i = i0 + threadIdx.x;
      const bool in = i < end;
      uint32_t k[9];
      if (in) load_biased(scalars, i, rc, k);
      for (int w = 0; w < ndigits; w++) {
        bool neg = false;
        const uint32_t d = in ? digit_of(k, w0 + w, c, neg) : 0u;
        const uint32_t bkt = d - 1;
        const uint32_t pos = wave_counter_add(cnt, (uint32_t)w, d != 0);
        if (d == 0) continue;
        if (FINE) atomicAdd(&fine_cnt[((uint32_t)w << fan_log) + (bkt >> fine_log)], 1u);
        if (WRITE) {
          rec_entry[pos] = i | (neg ? 0x80000000u : 0u);
          rec_bkt[pos] = (K)bkt;
        }
      }
    }
  } else
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    uint32_t k[9];
    load_biased(scalars, i, rc, k);
    for (int w = 0; w < ndigits; w++) {
      bool neg;
      const uint32_t d = digit_of(k, WIN ? w0 + w : w, c, neg);
      if (d == 0) continue;
      const uint32_t bkt = d - 1;
      const uint32_t q = WIN ? (uint32_t)w * ppw + (w == w_top ? (i & (ppw - 1u)) : (bkt >> nb_log)) : bkt >> nb_log;
      const uint32_t pos = atomicAdd(&cnt[q], 1u);
      if (FINE) atomicAdd(&fine_cnt[(q << fan_log) + ((bkt & (nb - 1u)) >> fine_log)], 1u);
      if (WRITE) {
        rec_entry[pos] = (WIN ? i : (uint32_t)w * n + i) | (neg ? 0x80000000u : 0u);
        rec_bkt[pos] = (K)(bkt & (nb - 1u));
      }
    }
  }
  if (!WRITE) {
    __syncthreads();
    if (threadIdx.x < P) blkcnt[blockIdx.x * P + threadIdx.x] = cnt[threadIdx.x];
    if (FINE)
      for (uint32_t f = threadIdx.x; f < (P << fan_log); f += blockDim.x) {
        const uint32_t v = fine_cnt[f];
        if (v) atomicAdd(&fine_tot[f], v);
      }
  }
}

// The write pass of the WINDOWED record pre-pass with an LDS stage (big plans and the two-level 16-bit windows: up to 16
// digits, P <= PART_MAX groups).  The direct form (k_part_pass<true, true>) lets every lane store its record halves wherever its
// group's cursor points: WRITE_SIZE 12.7 GB for 7 GB of records at 2^26 terms.  Here a batch of 1 024 scalars (13 312 records
// at c = 20) is ranked per group in LDS, laid out group by group in the stage and written out by consecutive lanes: a
// group's records of the batch leave as one run (64 records on average at c = 20, 1 024 at c = 16).  Slots: the
// (block, group) ranges k_part_scan assigned, as for the direct form.
constexpr uint32_t WSTAGE_MAXD = 16;
template <class K>
__global__ void __launch_bounds__(1024)
k_part_write_staged(const uint32_t* __restrict__ scalars, uint32_t n, int c, int ndigits, uint32_t nb, int nb_log, uint32_t P, uint32_t chunk,
                    RecodeConst rc, const uint32_t* __restrict__ blkcnt, uint32_t* __restrict__ rec_entry, K* __restrict__ rec_bkt, int w0,
                    int w_top_pos) {
  __shared__ uint32_t cursor[PART_MAX];  // next global record slot of (this block, group)
  __shared__ uint32_t bcnt[PART_MAX];    // records of the batch per group
  __shared__ uint32_t bbase[PART_MAX];   // ... their first stage slot
  __shared__ uint32_t gbase[PART_MAX];   // ... their first global slot
  __shared__ uint32_t s_total;
  extern __shared__ uint32_t wstage[];   // [1024 * ndigits] entries, then as many (group << 16 | bucket)
  const uint32_t stage_n = 1024u * (uint32_t)ndigits, tid = threadIdx.x;
  uint32_t* const s_entry = wstage;
  uint32_t* const s_key = wstage + stage_n;
  const uint32_t ppw = (1u << (c - 1)) >> nb_log;
  const int w_top = w_top_pos - w0;
  if (tid < PART_MAX) cursor[tid] = tid < P ? blkcnt[blockIdx.x * P + tid] : 0u;
  const uint32_t beg = blockIdx.x * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  // (the next batch's scalars are requested as soon as this batch's digits are extracted: the loads run under the staging and the write-out)
  uint32_t nk[9];
  if (beg + tid < end) load_biased(scalars, beg + tid, rc, nk);
  for (uint32_t i0 = beg; i0 < end; i0 += 1024) {
    if (tid < PART_MAX) bcnt[tid] = 0;
    __syncthreads();
    const uint32_t i = i0 + tid;
    const bool in = i < end;
    uint32_t ent[WSTAGE_MAXD], key[WSTAGE_MAXD], rk[WSTAGE_MAXD];
    uint32_t k[9];
#pragma unroll
    for (int j = 0; j < 9; j++) k[j] = nk[j];
    if (i + 1024 < end) load_biased(scalars, i + 1024, rc, nk);
#pragma unroll
    for (int w = 0; w < (int)WSTAGE_MAXD; w++) {
      key[w] = 0xffffffffu;
      if (w < ndigits) {  // (uniform)
        bool neg = false;
        const uint32_t d = in ? digit_of(k, w0 + w, c, neg) : 0u;
        const uint32_t bkt = d - 1;
        const uint32_t q = (uint32_t)w * ppw + (w == w_top ? (i & (ppw - 1u)) : (bkt >> nb_log));
        // one group per window: the wave's lanes name the same counter (aggregated update); 16 groups per window: plain atomics
        const uint32_t r = ppw == 1 ? wave_counter_add(bcnt, (uint32_t)w, d != 0) : (d != 0 ? atomicAdd(&bcnt[q], 1u) : 0u);
        if (d != 0) {
          ent[w] = i | (neg ? 0x80000000u : 0u);
          key[w] = (q << 16) | (bkt & (nb - 1u));
          rk[w] = r;
        }
      }
    }
    __syncthreads();
    if (tid < 64) {  // one wave, four groups per lane: stage and global bases of the batch's groups
      uint32_t v[4], sum = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        v[j] = bcnt[4 * tid + j];
        sum += v[j];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)tid >= off) incl += t;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t q = 4 * tid + j;
        bbase[q] = run;
        gbase[q] = cursor[q];
        cursor[q] += v[j];
        run += v[j];
      }
      if (tid == 63) s_total = incl;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < (int)WSTAGE_MAXD; w++)
      if (key[w] != 0xffffffffu) {
        const uint32_t slot = bbase[key[w] >> 16] + rk[w];
        s_entry[slot] = ent[w];
        s_key[slot] = key[w];
      }
    __syncthreads();
    const uint32_t total = s_total;
    for (uint32_t sl = tid; sl < total; sl += 1024) {
      const uint32_t kk = s_key[sl], q = kk >> 16;
      const uint32_t dst = gbase[q] + (sl - bbase[q]);
      rec_entry[dst] = s_entry[sl];
      rec_bkt[dst] = (K)(kk & 0xffffu);
    }
    __syncthreads();
  }
}

// blkcnt[blk][q] -> first record slot of (blk, q); part_total[q] = records of partition q
__global__ void __launch_bounds__(PART_MAX)
k_part_scan(uint32_t* __restrict__ blkcnt, uint32_t nblk, uint32_t P, uint32_t* __restrict__ part_total) {
  __shared__ uint32_t tot[PART_MAX];
  const uint32_t q = threadIdx.x;
  uint32_t s = 0;
  if (q < P)
    for (uint32_t b = 0; b < nblk; b++) s += blkcnt[b * P + q];
  tot[q] = (q < P) ? s : 0u;
  __syncthreads();
  if (q < P) {
    uint32_t run = 0;
    for (uint32_t j = 0; j < q; j++) run += tot[j];
    for (uint32_t b = 0; b < nblk; b++) {
      const uint32_t v = blkcnt[b * P + q];
      blkcnt[b * P + q] = run;
      run += v;
    }
    part_total[q] = s;
  }
}

template <bool SCATTER>
__global__ void __launch_bounds__(1024)
k_bucket_pass_rec(const uint32_t* __restrict__ rec_entry, const uint32_t* __restrict__ rec_bkt,
                  const uint32_t* __restrict__ part_total, uint32_t nb, uint32_t* __restrict__ blockhist,
                  uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t hist[];
  const uint32_t ch = blockIdx.x, q = blockIdx.y, nch = gridDim.x;
  uint32_t* gh = blockhist + ((size_t)q * nch + ch) * nb;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = SCATTER ? gh[b] : 0u;
  uint32_t pbase = 0;
  for (uint32_t j = 0; j < q; j++) pbase += part_total[j];
  const uint32_t ptot = part_total[q];
  const uint32_t len = (ptot + nch - 1) / nch;
  const uint32_t lo = ch * len;
  const uint32_t hi = (lo + len < ptot) ? lo + len : ptot;
  __syncthreads();
  for (uint32_t r = lo + threadIdx.x; r < hi; r += blockDim.x) {
    const uint32_t local = rec_bkt[pbase + r];
    if (SCATTER) {
      const uint32_t pos = atomicAdd(&hist[local], 1u);
      sorted[pos] = rec_entry[pbase + r];
    } else {
      atomicAdd(&hist[local], 1u);
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) gh[b] = hist[b];
  }
}

// ---- fine-partition sort (shared mode, more than 2^15 buckets) ---------------------------------
// The (chunk, 2^15-bucket partition) tiles above scatter 4-byte entries at random into a partition's slice of
// sorted[]: a bucket's run of ~26 entries receives ~1.6 entries from each of 16 tiles, so nearly every write is a
// partial line (WRITE_SIZE 8x the payload in the round-2 counters, 97 % of wave cycles waiting).  Here the records
// are grouped by FINE partition instead (2^10 consecutive buckets, ~27 k records at N = 2^20) and ONE workgroup owns
// a fine partition end to end: LDS histogram of its records -> prefix sum -> count[] / begin[] of its buckets ->
// scatter into its own contiguous slice of sorted[] (106 KB, staged in LDS: the runs of a bucket are completed by one workgroup
// within microseconds, so the lines are written once).  No per-tile histograms in HBM, no separate totals / scan /
// bases kernels for this mode.
constexpr uint32_t FINE_NB = 1u << FINE_LOG;  // (FINE_LOG = 10: above, with the record pre-pass)
constexpr uint32_t FINE_MAX_PARTS = 2048;    // 2^22 buckets (c = 23)
constexpr uint32_t FINE_BIG_PARTS = 32768;   // big windowed plans (run_windowed_big): 13 windows x 2^19 buckets in fine partitions of 2^8 = 26 624
constexpr uint32_t FPART_BLOCKS = 512;       // blocks of the record pre-pass

// records grouped by fine partition: WRITE = false counts per (block, partition), WRITE = true writes
// rec_entry (table index | sign) and rec_bkt (bucket inside the fine partition) at the slots k_fpart_scan assigned
template <bool WRITE>
__global__ void __launch_bounds__(1024)
k_fpart_pass(const uint32_t* __restrict__ scalars, uint32_t n, int c, int ndigits, uint32_t NP, uint32_t chunk, RecodeConst rc,
             uint32_t* __restrict__ blkcnt, const uint32_t* __restrict__ fpart, uint32_t* __restrict__ rec_entry,
             uint32_t* __restrict__ rec_bkt) {
  extern __shared__ uint32_t cnt[];  // NP counters / cursors
  // blkcnt is partition-major ([q][block], stride FPART_BLOCKS) so that k_fpart_scan's waves read a partition's row coalesced
  for (uint32_t q = threadIdx.x; q < NP; q += blockDim.x)
    cnt[q] = WRITE ? fpart[q] + blkcnt[(size_t)q * FPART_BLOCKS + blockIdx.x] : 0u;
  __syncthreads();
  const uint32_t beg = blockIdx.x * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    uint32_t k[9];
    load_biased(scalars, i, rc, k);
    for (int w = 0; w < ndigits; w++) {
      bool neg;
      const uint32_t d = digit_of(k, w, c, neg);
      if (d == 0) continue;
      const uint32_t bkt = d - 1;
      const uint32_t pos = atomicAdd(&cnt[bkt >> FINE_LOG], 1u);
      if (WRITE) {
        rec_entry[pos] = ((uint32_t)w * n + i) | (neg ? 0x80000000u : 0u);
        rec_bkt[pos] = bkt & (FINE_NB - 1u);
      }
    }
  }
  if (!WRITE) {
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < NP; q += blockDim.x) blkcnt[(size_t)q * FPART_BLOCKS + blockIdx.x] = cnt[q];
  }
}

// The write pass of the record pre-pass with an LDS stage (NP <= 512, ndigits <= 16): the direct form above lets every
// lane store its 4-byte record halves wherever its LDS cursor points -- each 64-byte line of a (block, partition) run is
// hit by 16 separate stores over the block's lifetime and half of them reach HBM as partial lines (WRITE_SIZE 678 MB for
// 109 MB of records).  Here a batch of 1024 scalars (<= 16 K records) is ranked per partition in LDS, laid out partition
// by partition in the stage, and written out by consecutive lanes: a partition's records of the batch leave as one run.
constexpr uint32_t FPASS_MAXD = 16;                       // digits per scalar the staged pass supports
// NPMAX = 512 (plans of up to 2^19 buckets: the prover at N <= 2^20) or 2048 (up to 2^21 buckets: N = 2^21, 2^22); the stage
// holds 1024 x ndigits records (dynamic LDS: 128 KB at 16 digits, 96 KB at the 12 digits of the big plans).
template <int NPMAX>
__global__ void __launch_bounds__(1024)
k_fpart_write_staged(const uint32_t* __restrict__ scalars, uint32_t n, int c, int ndigits, uint32_t NP, uint32_t chunk, RecodeConst rc,
                     const uint32_t* __restrict__ blkcnt, const uint32_t* __restrict__ fpart, uint32_t* __restrict__ rec_entry,
                     uint32_t* __restrict__ rec_bkt) {
  __shared__ uint32_t cursor[NPMAX];  // next global record slot of (this block, partition)
  __shared__ uint32_t bcnt[NPMAX];    // records of the batch per partition, then their first stage slot (exclusive prefix)
  __shared__ uint32_t gbase[NPMAX];   // global slot of the batch's first record of the partition
  __shared__ uint32_t part[1024];
  extern __shared__ uint32_t stg[];  // [1024 * ndigits] entries, then as many (partition << FINE_LOG | bucket)
  const uint32_t stage_n = 1024u * (uint32_t)ndigits;
  uint32_t* const s_entry = stg;
  uint32_t* const s_qb = stg + stage_n;
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t PER = NPMAX > 1024 ? NPMAX / 1024 : 1;  // partitions per thread in the scan
  for (uint32_t q = tid; q < (uint32_t)NPMAX; q += 1024) cursor[q] = q < NP ? fpart[q] + blkcnt[(size_t)q * FPART_BLOCKS + blockIdx.x] : 0u;
  const uint32_t beg = blockIdx.x * chunk;
  const uint32_t end = (beg + chunk < n) ? beg + chunk : n;
  for (uint32_t i0 = beg; i0 < end; i0 += 1024) {
    for (uint32_t q = tid; q < (uint32_t)NPMAX; q += 1024) bcnt[q] = 0;
    __syncthreads();
    const uint32_t i = i0 + tid;
    uint32_t ent[FPASS_MAXD], qb[FPASS_MAXD], rk[FPASS_MAXD];
    if (i < end) {
      uint32_t k[9];
      load_biased(scalars, i, rc, k);
#pragma unroll
      for (int w = 0; w < (int)FPASS_MAXD; w++) {
        qb[w] = 0xffffffffu;
        if (w < ndigits) {
          bool neg;
          const uint32_t d = digit_of(k, w, c, neg);
          if (d) {
            const uint32_t bkt = d - 1;
            ent[w] = ((uint32_t)w * n + i) | (neg ? 0x80000000u : 0u);
            qb[w] = bkt;  // partition = bkt >> FINE_LOG, bucket = low bits
            rk[w] = atomicAdd(&bcnt[bkt >> FINE_LOG], 1u);
          }
        }
      }
    } else {
#pragma unroll
      for (int w = 0; w < (int)FPASS_MAXD; w++) qb[w] = 0xffffffffu;
    }
    __syncthreads();
    // exclusive prefix of bcnt over the partitions (thread t owns partitions [t * PER, (t + 1) * PER)), batch bases from the cursors
    uint32_t mine[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t q = tid * PER + k;
      mine[k] = q < (uint32_t)NPMAX ? bcnt[q] : 0u;
      sum += mine[k];
    }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    uint32_t run = part[tid] - sum;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t q = tid * PER + k;
      if (q < (uint32_t)NPMAX) {
        bcnt[q] = run;  // first stage slot of the partition
        gbase[q] = cursor[q];
        cursor[q] += mine[k];
        run += mine[k];
      }
    }
    const uint32_t total = part[1023];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < (int)FPASS_MAXD; w++)
      if (qb[w] != 0xffffffffu) {
        const uint32_t slot = bcnt[qb[w] >> FINE_LOG] + rk[w];
        s_entry[slot] = ent[w];
        s_qb[slot] = qb[w];
      }
    __syncthreads();
    for (uint32_t sl = tid; sl < total; sl += 1024) {
      const uint32_t v = s_qb[sl];
      const uint32_t q = v >> FINE_LOG;
      const uint32_t dst = gbase[q] + (sl - bcnt[q]);
      rec_entry[dst] = s_entry[sl];
      rec_bkt[dst] = v & (FINE_NB - 1u);
    }
    __syncthreads();
  }
}

// one WAVE per fine partition: exclusive prefix of its row blkcnt[q][0 .. nblk) in place (the slot of (block, q) relative
// to the partition's first record) and the row total -> fpart[NP + q]
__global__ void __launch_bounds__(64)
k_fpart_scan_rows(uint32_t* __restrict__ blkcnt, uint32_t nblk, uint32_t NP, uint32_t* __restrict__ fpart) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  uint32_t* row = blkcnt + (size_t)q * FPART_BLOCKS;
  constexpr uint32_t PER = FPART_BLOCKS / 64;
  uint32_t v[PER], s = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t b = lane * PER + k;
    v[k] = b < nblk ? row[b] : 0u;
    s += v[k];
  }
  uint32_t incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if ((int)lane >= off) incl += t;
  }
  uint32_t run = incl - s;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t b = lane * PER + k;
    if (b < nblk) row[b] = run;
    run += v[k];
  }
  if (lane == 63) fpart[NP + q] = incl;
}
// one workgroup: fpart[q] = first record slot of partition q (exclusive prefix of the totals fpart[NP + q])
// It also leaves the sort's record count where the other sort paths leave their per-partition totals: part_total[0] = all
// records, part_total[1 .. P) = 0 (the prover's k_entries_to_host sums the P words: the non-zero digits of the assignment)
__global__ void __launch_bounds__(1024)
k_fpart_scan_base(uint32_t NP, uint32_t* __restrict__ fpart, uint32_t* __restrict__ part_total, uint32_t P,
                  uint32_t* __restrict__ big_list = nullptr) {
  if (big_list && threadIdx.x == 0) big_list[0] = 0;  // (k_fine_cursors, the next kernel of the stream, fills the list)
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (NP + 1023u) / 1024u;
  uint32_t s = 0;
  for (uint32_t q = tid * per; q < (tid + 1) * per && q < NP; q++) s += fpart[NP + q];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0u;
  for (uint32_t q = tid * per; q < (tid + 1) * per && q < NP; q++) {
    fpart[q] = run;
    run += fpart[NP + q];
  }
  if (part_total && tid < P) part_total[tid] = tid == 0 ? part[1023] : 0u;
}

// one workgroup per fine partition: histogram -> scan -> count / begin -> scatter (see above).
// The scatter is STAGED in LDS: cursors hand out final positions, the entry goes to stage[position - round base] and the
// stage is flushed to sorted[] as whole lines, so the 4-byte entries of a bucket's run no longer reach HBM one partial
// line at a time (WRITE_SIZE 6.5x the payload for the direct scatter).  A partition bigger than the stage (the fine
// partitions that also receive the partial top digit hold ~3.6x the mean; witness-like scalars can put 20 % of all
// entries into one bucket) is flushed in rounds of FINE_ROUND positions: bucket b belongs to the round its first
// position falls into, a run that overshoots the stage's slack is written directly.
constexpr uint32_t FINE_STAGE = 36864;   // entries the LDS stage holds (147 456 B)
constexpr uint32_t FINE_ROUND = 32768;   // positions per round; FINE_STAGE - FINE_ROUND = slack for a straddling run
constexpr uint32_t FSORT_U = 4;     // records in flight per thread in the round form of k_fpart_sort
constexpr uint32_t FSORT_RPT = 36;  // records per thread of its register form: partitions of up to 36 864 records (= FINE_STAGE)
// fine_log: log2 of the buckets per fine partition (<= FINE_LOG: one counter per thread); bucket ids are q << fine_log | local.
// A partition of at most FSORT_RPT x 1024 records takes the REGISTER form: every thread loads its records once, all loads in
// flight together (72 per thread), and keeps them through histogram, scan and the scatter into the stage -- one pass over HBM
// where the round form below reads the bucket ids once per pass and round with a handful of loads in flight (16 waves per CU
// beside a 147 KB stage hide no latency: 58 G records/s at 131 072 records per partition, 75 G at 26 000).
template <class K>
__global__ void __launch_bounds__(1024)
k_fpart_sort(const uint32_t* __restrict__ rec_entry, const K* __restrict__ rec_bkt, const uint32_t* __restrict__ fpart,
             uint32_t NP, uint32_t* __restrict__ count, uint32_t* __restrict__ begin, uint32_t* __restrict__ sorted, int staged,
             int fine_log, const uint32_t* __restrict__ bigflag = nullptr) {
  if (bigflag && bigflag[blockIdx.x]) return;  // an oversized partition: sorted by the k_big_* kernels, 64 workgroups each
  static_assert(FINE_NB == 1024, "one counter per thread");
  static_assert(FSORT_RPT * 1024 <= FINE_STAGE, "the register form stages a whole partition");
  __shared__ uint32_t hist[FINE_NB];  // counts, then cursors (positions relative to the partition)
  __shared__ uint32_t beg[FINE_NB];   // first position of the bucket, relative to the partition
  __shared__ uint32_t part[1024];
  extern __shared__ uint32_t stage[];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t pbase = fpart[q], ptot = fpart[NP + q];
  const uint32_t ptot_up = (ptot + 1023u) & ~1023u;  // whole waves walk the records: the aggregated counter updates need every lane
  const bool in_regs = staged && ptot <= FSORT_RPT * 1024;  // workgroup-uniform
  uint32_t kb[FSORT_RPT], ke[FSORT_RPT];
  hist[tid] = 0;
  if (in_regs) {
#pragma unroll
    for (uint32_t k = 0; k < FSORT_RPT; k++) {
      const uint32_t r = k * 1024 + tid;
      kb[k] = r < ptot ? rec_bkt[pbase + r] : 0xffffffffu;
      ke[k] = r < ptot ? rec_entry[pbase + r] : 0u;
    }
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (uint32_t k = 0; k < FSORT_RPT; k++)
      if (k * 1024 < ptot_up) (void)wave_counter_add(hist, kb[k] != 0xffffffffu ? kb[k] : 0u, kb[k] != 0xffffffffu);
  } else {
    for (uint32_t r0 = tid; r0 < ptot_up; r0 += FSORT_U * 1024) {
      uint32_t b[FSORT_U];
#pragma unroll
      for (uint32_t k = 0; k < FSORT_U; k++) b[k] = (r0 + k * 1024 < ptot) ? rec_bkt[pbase + r0 + k * 1024] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < FSORT_U; k++)
        if (r0 + k * 1024 < ptot_up) (void)wave_counter_add(hist, b[k], r0 + k * 1024 < ptot);
    }
  }
  __syncthreads();
  const uint32_t c0 = hist[tid];
  part[tid] = c0;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const uint32_t ex = tid ? part[tid - 1] : 0u;
  if (tid < (1u << fine_log)) {
    const size_t g = ((size_t)q << fine_log) + tid;
    count[g] = c0;
    begin[g] = pbase + ex;
  }
  beg[tid] = ex;
  hist[tid] = ex;  // cursor
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (uint32_t k = 0; k < FSORT_RPT; k++)
      if (k * 1024 < ptot_up) {
        const bool valid = kb[k] != 0xffffffffu;
        const uint32_t pos = wave_counter_add(hist, valid ? kb[k] : 0u, valid);
        if (valid) stage[pos] = ke[k];
      }
    __syncthreads();
    for (uint32_t i = tid; i < ptot; i += 1024) sorted[pbase + i] = stage[i];
    return;
  }
  if (!staged) {
    for (uint32_t r = tid; r < ptot_up; r += 1024) {
      const bool valid = r < ptot;
      const uint32_t pos = wave_counter_add(hist, valid ? rec_bkt[pbase + r] : 0u, valid);
      if (valid) sorted[pbase + pos] = rec_entry[pbase + r];
    }
    return;
  }
  const uint32_t rounds = (ptot + FINE_ROUND - 1) / FINE_ROUND;
  uint32_t start = 0;  // first position this round owns = end of the previous round's last run (>= lo)
  for (uint32_t rd = 0; rd < rounds;) {
    const uint32_t lo = rd * FINE_ROUND;
    for (uint32_t r0 = tid; r0 < ptot_up; r0 += FSORT_U * 1024) {
      uint32_t b[FSORT_U], e[FSORT_U];
      bool valid[FSORT_U];
#pragma unroll
      for (uint32_t k = 0; k < FSORT_U; k++) b[k] = (r0 + k * 1024 < ptot) ? rec_bkt[pbase + r0 + k * 1024] : 0u;
#pragma unroll
      for (uint32_t k = 0; k < FSORT_U; k++) {
        valid[k] = r0 + k * 1024 < ptot && (rounds == 1 || beg[b[k]] / FINE_ROUND == rd);
        e[k] = valid[k] ? rec_entry[pbase + r0 + k * 1024] : 0u;
      }
#pragma unroll
      for (uint32_t k = 0; k < FSORT_U; k++) {
        if (r0 + k * 1024 >= ptot_up) continue;  // (wave-uniform: ptot_up is a multiple of 1024)
        const uint32_t pos = wave_counter_add(hist, b[k], valid[k]);
        if (!valid[k]) continue;
        if (pos - lo < FINE_STAGE) stage[pos - lo] = e[k];
        else sorted[pbase + pos] = e[k];  // a run longer than the slack: the rest goes out directly
      }
    }
    __syncthreads();
    // positions [lo, hi) were staged: hi = end of the last bucket of this round, capped by the stage
    uint32_t hi = ptot, next_rd = rounds;
    if (rd + 1 < rounds) {
      // first bucket of a later round = first b with beg[b] >= (rd + 1) * FINE_ROUND; its beg is the end of this round
      // (found by every thread from the monotone beg[] with a binary search: 10 steps)
      uint32_t l = 0, h = FINE_NB;
      while (l < h) {
        const uint32_t m = (l + h) >> 1;
        if (beg[m] >= (rd + 1) * FINE_ROUND) h = m;
        else l = m + 1;
      }
      hi = (l < FINE_NB) ? beg[l] : ptot;
      // the next round that OWNS a bucket: a run of 600 000 positions spans 18 rounds in which no bucket begins, and each
      // of them used to walk all the partition's records for nothing
      next_rd = (l < FINE_NB && beg[l] < ptot) ? beg[l] / FINE_ROUND : rounds;
    }
    uint32_t top = hi - lo;
    if (top > FINE_STAGE) top = FINE_STAGE;
    for (uint32_t k = start - lo + tid; k < top; k += 1024) sorted[pbase + lo + k] = stage[k];
    start = hi;
    rd = next_rd;
    __syncthreads();
  }
}

// ---- big windowed plans: records of a 2^15-bucket group -> records grouped by FINE partition ------------------------------
// The (chunk, group) tiles of k_bucket_pass_rec scatter 4-byte entries at random into the group's 16 MB slice of sorted[]: at
// 2^26 terms 872 M entries cost 26.9 GB of HBM writes for 3.5 GB of payload and 20.5 of the sort's 28.9 ms (round-5 counters,
// profiles/r05/pmc_summary_msm26_steps1.json).  Instead the group's records are split once more, into its 32 .. 128 fine
// partitions of 2^10 .. 2^8 buckets (as many as keep a fine partition within one LDS stage: 32 768 records on average at 2^26
// terms), and k_fpart_sort finishes each fine partition inside one workgroup -- the scheme of the prover's sort, with one more
// level because a record pre-pass straight into 26 624 fine partitions would write single records.
// A workgroup takes batches of SPLIT_B records of its group: ranks them per fine partition (wave-aggregated: the lanes of a wave
// that name the same partition are found with ballots and take their LDS counter once), reserves the batch's room in every
// fine partition with ONE global atomic each, lays the batch out partition by partition in LDS and writes it out as runs
// (SPLIT_B / fan = 64 .. 256 records on average).  The order of a fine partition's records depends on which batch reserved first:
// the bucket sums do not (and every caller's result is canonical bytes).
constexpr uint32_t SPLIT_B = 8192;  // records per batch: 8 per thread, 64 KB of stage
constexpr uint32_t SPLIT_MAX_FAN = 256;  // 2^15-bucket groups cut into fine partitions of 2^7 buckets
__device__ __forceinline__ uint32_t wave_bin_rank(uint32_t* ctr, uint32_t f, bool valid, int fan_log) {
  uint64_t m = __ballot(valid);
  for (int bit = 0; bit < fan_log; bit++) {
    const bool one = ((f >> bit) & 1u) != 0;
    const uint64_t bl = __ballot(valid && one);
    m &= one ? bl : ~bl;
  }
  if (!valid) m = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const int leader = m ? __ffsll((unsigned long long)m) - 1 : (int)lane;
  uint32_t base = 0;
  if (valid && (int)lane == leader) base = atomicAdd(&ctr[f], (uint32_t)__popcll(m));
  base = (uint32_t)__shfl((int)base, leader);
  return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
template <class K>
__global__ void __launch_bounds__(1024)
k_rec_split(const uint32_t* __restrict__ a_entry, const K* __restrict__ a_bkt, const uint32_t* __restrict__ part_total,
            uint32_t* __restrict__ cursor, uint32_t* __restrict__ b_entry, K* __restrict__ b_bkt, int fan_log, int fine_log, int plain_rank) {
  extern __shared__ uint32_t split_stage[];  // SPLIT_B entries, then SPLIT_B keys
  __shared__ uint32_t cnt[SPLIT_MAX_FAN], base[SPLIT_MAX_FAN], gb[SPLIT_MAX_FAN];
  uint32_t* const s_entry = split_stage;
  uint32_t* const s_key = split_stage + SPLIT_B;
  const uint32_t tid = threadIdx.x, q = blockIdx.y, fan = 1u << fan_log;
  uint32_t pbase = 0;
  for (uint32_t j = 0; j < q; j++) pbase += part_total[j];
  const uint32_t ptot = part_total[q];
  constexpr uint32_t PER = SPLIT_B / 1024;
  // (the next batch's records are requested as soon as this batch's sit in the stage: its loads run under the write-out)
  uint32_t nkey[PER], nent[PER];
  auto fetch = [&](uint32_t b0) {
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      const uint32_t r = b0 + k * 1024 + tid;
      nkey[k] = r < ptot ? (uint32_t)a_bkt[pbase + r] : 0xffffffffu;
      nent[k] = r < ptot ? a_entry[pbase + r] : 0u;
    }
  };
  if (blockIdx.x * SPLIT_B < ptot) fetch(blockIdx.x * SPLIT_B);
  for (uint32_t b0 = blockIdx.x * SPLIT_B; b0 < ptot; b0 += gridDim.x * SPLIT_B) {
    if (tid < fan) cnt[tid] = 0;
    __syncthreads();
    uint32_t key[PER], ent[PER], rk[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
      key[k] = nkey[k];
      ent[k] = nent[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++)
      rk[k] = plain_rank ? (key[k] != 0xffffffffu ? atomicAdd(&cnt[key[k] >> fine_log], 1u) : 0u)
                         : wave_bin_rank(cnt, key[k] >> fine_log, key[k] != 0xffffffffu, fan_log);
    __syncthreads();
    if (tid < 64) {  // one wave (four partitions per lane): exclusive prefix of the batch's counts, and the batch's room in every fine partition
      uint32_t v[4], sum = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t f = 4 * tid + j;
        v[j] = f < fan ? cnt[f] : 0u;
        sum += v[j];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)tid >= off) incl += t;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t f = 4 * tid + j;
        if (f < fan) {
          base[f] = run;
          gb[f] = v[j] ? atomicAdd(&cursor[(q << fan_log) + f], v[j]) : 0u;
          run += v[j];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++)
      if (key[k] != 0xffffffffu) {
        const uint32_t slot = base[key[k] >> fine_log] + rk[k];
        s_entry[slot] = ent[k];
        s_key[slot] = key[k];
      }
    if (b0 + gridDim.x * SPLIT_B < ptot) fetch(b0 + gridDim.x * SPLIT_B);
    __syncthreads();
    const uint32_t total = ptot - b0 < SPLIT_B ? ptot - b0 : SPLIT_B;
    for (uint32_t sl = tid; sl < total; sl += 1024) {
      const uint32_t kk = s_key[sl], f = kk >> fine_log;
      const uint32_t dst = gb[f] + (sl - base[f]);
      b_entry[dst] = s_entry[sl];
      b_bkt[dst] = (K)(kk & ((1u << fine_log) - 1u));
    }
    __syncthreads();
  }
}
// Oversized fine partitions.  One workgroup per fine partition is the right grain while the digits are spread out; a witness of
// bits and small values is not: a fifth of all scalars equal to one puts 13 M records of a 2^26-term MSM into ONE bucket, and
// the workgroup owning it would walk them three times while 26 623 others finished in microseconds.  Partitions above
// BIG_THR records (four times the largest mean the planner allows) are listed (at most BIG_MAX) and sorted by BIG_NCH
// workgroups each in three small kernels -- per-chunk LDS histograms, a scan over chunks and buckets that also writes count[] /
// begin[], and a scatter with LDS cursors (a heavy bucket's entries leave as whole runs: the lanes of a wave that name the same
// bucket take consecutive positions); k_fpart_sort skips them.  With uniform digits the list is empty and the three launches
// return at once.
constexpr uint32_t BIG_THR = 131072, BIG_MAX = 32, BIG_NCH = 64;
// fpart[2 NP + q] = fpart[q]: the cursors k_rec_split advances; fpart[3 NP + q] = 1 + list index of an oversized partition, or 0
__global__ void __launch_bounds__(256)
k_fine_cursors(uint32_t* __restrict__ fpart, uint32_t NP, uint32_t* __restrict__ big_list, uint32_t max_big) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= NP) return;
  fpart[2 * NP + q] = fpart[q];
  uint32_t flag = 0;
  if (fpart[NP + q] > BIG_THR) {
    const uint32_t j = atomicAdd(&big_list[0], 1u);
    if (j < max_big) {  // (= the y extent of the k_big_* grids: at most BIG_MAX)
      big_list[1 + j] = q;
      flag = 1 + j;
    }
  }
  fpart[3 * NP + q] = flag;
}
// chunk ch of listed partition j: records [lo, hi) of the partition (whole waves walk them: the aggregated counters need every lane)
__device__ __forceinline__ bool big_slice(const uint32_t* fpart, uint32_t NP, const uint32_t* big_list, uint32_t& pbase, uint32_t& lo,
                                          uint32_t& hi, uint32_t& steps) {
  const uint32_t j = blockIdx.y, nbig = big_list[0] < gridDim.y ? big_list[0] : gridDim.y;
  if (j >= nbig) return false;
  const uint32_t q = big_list[1 + j];
  pbase = fpart[q];
  const uint32_t ptot = fpart[NP + q];
  const uint32_t per = (((ptot + BIG_NCH - 1) / BIG_NCH) + 1023u) & ~1023u;
  lo = blockIdx.x * per;
  hi = lo + per < ptot ? lo + per : ptot;
  steps = per / 1024;
  return true;
}
template <class K>
__global__ void __launch_bounds__(1024)
k_big_hist(const K* __restrict__ rec_bkt, const uint32_t* __restrict__ fpart, uint32_t NP, const uint32_t* __restrict__ big_list,
           uint32_t* __restrict__ bighist) {
  __shared__ uint32_t hist[FINE_NB];
  uint32_t pbase, lo, hi, steps;
  if (!big_slice(fpart, NP, big_list, pbase, lo, hi, steps)) return;
  const uint32_t tid = threadIdx.x;
  hist[tid] = 0;
  __syncthreads();
  for (uint32_t k0 = 0; k0 < steps; k0 += 4) {
    uint32_t b[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t r = lo + (k0 + k) * 1024 + tid;
      b[k] = (k0 + k < steps && r < hi) ? (uint32_t)rec_bkt[pbase + r] : 0xffffffffu;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (k0 + k < steps) (void)wave_counter_add(hist, b[k] != 0xffffffffu ? b[k] : 0u, b[k] != 0xffffffffu);
  }
  __syncthreads();
  bighist[((size_t)blockIdx.y * BIG_NCH + blockIdx.x) * FINE_NB + tid] = hist[tid];
}
// one workgroup per listed partition, thread = bucket: chunk counts -> absolute first positions per (chunk, bucket); count[] / begin[]
__global__ void __launch_bounds__(1024)
k_big_scan(const uint32_t* __restrict__ fpart, uint32_t NP, const uint32_t* __restrict__ big_list, uint32_t* __restrict__ bighist,
           uint32_t* __restrict__ count, uint32_t* __restrict__ begin, int fine_log) {
  __shared__ uint32_t part[1024];
  const uint32_t j = blockIdx.x, nbig = big_list[0] < gridDim.x ? big_list[0] : gridDim.x, tid = threadIdx.x;
  if (j >= nbig) return;
  const uint32_t q = big_list[1 + j], pbase = fpart[q];
  uint32_t* const row = bighist + (size_t)j * BIG_NCH * FINE_NB + tid;
  uint32_t run = 0;
  for (uint32_t ch = 0; ch < BIG_NCH; ch++) {
    const uint32_t v = row[(size_t)ch * FINE_NB];
    row[(size_t)ch * FINE_NB] = run;
    run += v;
  }
  part[tid] = run;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const uint32_t first = pbase + (tid ? part[tid - 1] : 0u);
  if (tid < (1u << fine_log)) {
    const size_t g = ((size_t)q << fine_log) + tid;
    count[g] = run;
    begin[g] = first;
  }
  for (uint32_t ch = 0; ch < BIG_NCH; ch++) row[(size_t)ch * FINE_NB] += first;
}
template <class K>
__global__ void __launch_bounds__(1024)
k_big_scatter(const uint32_t* __restrict__ rec_entry, const K* __restrict__ rec_bkt, const uint32_t* __restrict__ fpart, uint32_t NP,
              const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ bighist, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t cur[FINE_NB];
  uint32_t pbase, lo, hi, steps;
  if (!big_slice(fpart, NP, big_list, pbase, lo, hi, steps)) return;
  const uint32_t tid = threadIdx.x;
  cur[tid] = bighist[((size_t)blockIdx.y * BIG_NCH + blockIdx.x) * FINE_NB + tid];
  __syncthreads();
  for (uint32_t k0 = 0; k0 < steps; k0 += 4) {
    uint32_t b[4], e[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t r = lo + (k0 + k) * 1024 + tid;
      const bool in = k0 + k < steps && r < hi;
      b[k] = in ? (uint32_t)rec_bkt[pbase + r] : 0xffffffffu;
      e[k] = in ? rec_entry[pbase + r] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (k0 + k < steps) {
        const bool valid = b[k] != 0xffffffffu;
        const uint32_t pos = wave_counter_add(cur, valid ? b[k] : 0u, valid);
        if (valid) sorted[pos] = e[k];
      }
  }
}

__global__ void __launch_bounds__(1024)
k_part_totals(const uint32_t* __restrict__ count, uint32_t* __restrict__ part_total, uint32_t nb) {
  __shared__ uint32_t part[1024];
  uint32_t s = 0;
  for (uint32_t b = threadIdx.x; b < nb; b += 1024) s += count[blockIdx.x * nb + b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part_total[blockIdx.x] = part[0];
}

// Pass 2a: per (window, bucket) exclusive prefix over the chunks (in place) and the bucket total.
__global__ void __launch_bounds__(256)
k_bucket_totals(uint32_t* __restrict__ blockhist, uint32_t* __restrict__ count, uint32_t nb, uint32_t nch,
                uint32_t total_buckets) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_buckets) return;
  const uint32_t w = g / nb, b = g - w * nb;
  uint32_t run = 0;
  for (uint32_t ch = 0; ch < nch; ch++) {
    uint32_t* p = blockhist + ((size_t)w * nch + ch) * nb + b;
    const uint32_t v = *p;
    *p = run;
    run += v;
  }
  count[g] = run;
}

// Pass 2b: one workgroup per window: begin[w][b] = w*n + exclusive prefix of count[w][*].
__global__ void __launch_bounds__(1024)
k_window_scan(const uint32_t* __restrict__ count, uint32_t* __restrict__ begin, uint32_t nb, uint32_t n,
              const uint32_t* __restrict__ part_total) {
  __shared__ uint32_t part[1024];
  const uint32_t w = blockIdx.x, tid = threadIdx.x;
  // region of this window / partition in sorted[]: fixed w*n, or (shared mode) packed
  uint32_t region = w * n;
  if (part_total) {
    region = 0;
    for (uint32_t q = 0; q < w; q++) region += part_total[q];
  }
  const uint32_t per = (nb + 1023u) / 1024u;
  const uint32_t b0 = tid * per;
  const uint32_t b1 = (b0 + per < nb) ? b0 + per : nb;
  uint32_t s = 0;
  for (uint32_t b = b0; b < b1; b++) s += count[w * nb + b];
  part[tid] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = region + ((tid == 0) ? 0u : part[tid - 1]);
  for (uint32_t b = b0; b < b1; b++) {
    begin[w * nb + b] = run;
    run += count[w * nb + b];
  }
}

// Pass 2c: tile bases = bucket begin + prefix over chunks.
__global__ void __launch_bounds__(256)
k_bucket_bases(uint32_t* __restrict__ blockhist, const uint32_t* __restrict__ begin, uint32_t nb, uint32_t nch,
               uint32_t total) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint32_t b = g % nb;
  const uint32_t w = g / (nb * nch);
  blockhist[g] += begin[w * nb + b];
}

// Pass 2d: ALL buckets (every window / partition) ordered by descending load, so that
//   * the 64 lanes of a wave in k_accum own buckets of (nearly) equal size (a thread-per-bucket loop
//     otherwise runs at the pace of the fullest bucket: Poisson(32) loads give ~65 % lane utilisation);
//   * the dispatcher hands out the longest waves first (LPT).  Ordering each window on its own left a
//     saw-tooth: the chip holds a quarter of the waves at a time, and with heavy groups arriving in the
//     last quarter the accumulation ended on a long, thinly occupied tail (resident waves per SIMD 1.27 of
//     2 measured; a scheduling model of the same loads gives 0.79 -> 0.94 slot occupancy for the global order).
// Counting sort by the key (load >> shift) in three small kernels: per-block LDS histogram -> global bins,
// scan of the 258 bins, scatter with per-block reserved ranges.  Buckets above the heavy threshold go to a
// separate list that k_accum_heavy reduces with whole workgroups; they sort last (the accumulation skips them).
constexpr uint32_t ORDER_BINS = MSM_HEAVY + 2;
__device__ __forceinline__ uint32_t order_key(uint32_t cnt, uint32_t thr, int shift) {
  // 0 = heaviest light bucket ... MSM_HEAVY = empty; heavy buckets sort last (key MSM_HEAVY + 1)
  return cnt > thr ? MSM_HEAVY + 1 : MSM_HEAVY - (cnt >> shift);
}
// (No memset launches in this chain: a 4-byte hipMemsetAsync is a kernel of its own, and on a chip kept full by another
// stream's accumulation it waited 0.07 - 0.8 ms for a slot, ten times per proof.  Every block writes its own row of bins,
// the scan kernel sums the rows and also clears the heavy-bucket counter.)
constexpr uint32_t ORDER_MAX_BLOCKS = 256;
__global__ void __launch_bounds__(1024)
k_order_hist(const uint32_t* __restrict__ count, uint32_t* __restrict__ blockbins, uint32_t total, uint32_t thr, int shift) {
  __shared__ uint32_t bins[ORDER_BINS];
  for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += 1024) bins[i] = 0;
  __syncthreads();
  for (uint32_t b = blockIdx.x * 1024 + threadIdx.x; b < total; b += gridDim.x * 1024) atomicAdd(&bins[order_key(count[b], thr, shift)], 1u);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += 1024) blockbins[(size_t)(1 + blockIdx.x) * ORDER_BINS + i] = bins[i];
}
// one workgroup: gbins[i] (row 0) = exclusive prefix over the keys of the per-block rows' sums; *n_heavy = 0
__global__ void __launch_bounds__(512)
k_order_scan(uint32_t* __restrict__ blockbins, uint32_t nblocks, uint32_t* __restrict__ n_heavy) {
  static_assert(ORDER_BINS <= 512, "one key per thread");
  __shared__ uint32_t part[512];
  const uint32_t i = threadIdx.x;
  uint32_t v = 0;
  if (i < ORDER_BINS)
    for (uint32_t b = 0; b < nblocks; b++) v += blockbins[(size_t)(1 + b) * ORDER_BINS + i];
  part[i] = v;
  __syncthreads();
  for (uint32_t off = 1; off < 512; off <<= 1) {
    const uint32_t t = (i >= off) ? part[i - off] : 0u;
    __syncthreads();
    part[i] += t;
    __syncthreads();
  }
  if (i < ORDER_BINS) blockbins[i] = part[i] - v;
  if (i == 0) *n_heavy = 0;
}
__global__ void __launch_bounds__(1024)
k_order_scatter(const uint32_t* __restrict__ count, uint32_t* __restrict__ gbins, uint32_t* __restrict__ perm,
                uint32_t* __restrict__ heavy, uint32_t* __restrict__ n_heavy, uint32_t total, uint32_t thr, int shift) {
  __shared__ uint32_t bins[ORDER_BINS];
  for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += 1024) bins[i] = 0;
  __syncthreads();
  // every block owns one contiguous slice of the bucket array: count its keys, reserve a range per key, scatter
  const uint32_t per = (total + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per, hi = (lo + per < total) ? lo + per : total;
  for (uint32_t b = lo + threadIdx.x; b < hi; b += 1024) atomicAdd(&bins[order_key(count[b], thr, shift)], 1u);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += 1024) {
    const uint32_t c = bins[i];
    bins[i] = c ? atomicAdd(&gbins[i], c) : 0u;
  }
  __syncthreads();
  for (uint32_t b = lo + threadIdx.x; b < hi; b += 1024) {
    const uint32_t cnt = count[b];
    const uint32_t pos = atomicAdd(&bins[order_key(cnt, thr, shift)], 1u);
    perm[pos] = b;
    if (cnt > thr) heavy[atomicAdd(n_heavy, 1u)] = b;
  }
}

// records the launches of the three ordering kernels on `st`
static hipError_t bucket_order(const uint32_t* count, uint32_t* perm, uint32_t* heavy, uint32_t* order_bins, uint32_t total,
                               uint32_t thr, int shift, hipStream_t st) {
  uint32_t blocks = (total + 16383) / 16384;
  if (blocks > ORDER_MAX_BLOCKS) blocks = ORDER_MAX_BLOCKS;
  if (!blocks) blocks = 1;
  hipLaunchKernelGGL(k_order_hist, dim3(blocks), dim3(1024), 0, st, count, order_bins, total, thr, shift);
  hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(512), 0, st, order_bins, blocks, heavy);
  hipLaunchKernelGGL(k_order_scatter, dim3(blocks), dim3(1024), 0, st, count, order_bins, perm, heavy + 1, heavy, total, thr, shift);
  return hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------------------
static int pick_window(uint64_t n) {
  // signed digits: 2^(c-1) buckets per window; mean bucket load n / 2^(c-1)
  // kept >= ~16 so a thread-per-bucket accumulation has work per lane.
  if (n <= (1u << 8)) return 5;
  if (n <= (1u << 11)) return 8;
  if (n <= (1u << 14)) return 10;
  if (n <= (1u << 17)) return 13;
  // From 2^24 terms on a wider digit pays for its bigger bucket sets: c = 20 is 13 windows (12 full ones and a 15-bit top
  // window) instead of 16, i.e. 13 n instead of 16 n insertions, against 13 x 2^19 instead of 16 x 2^15 buckets to
  // reduce (~3.5 insertion-equivalents each): -10 % at 2^24, -17 % at 2^26.  c = 21 / 22 cost the same within 1 % with
  // two / four times the bucket memory.  A window of 2^19 buckets is sorted as 16 partitions of 2^15 (run_windowed_big).
  if (n >= (1ull << 24)) return MSM_BIG_WINDOW;
  return 16;  // 2^15 buckets = 128 KiB of LDS counters per (window, chunk) tile
}

static void plan_set_heavy(MsmPlan& p, uint64_t items) {
  const uint64_t buckets = (uint64_t)p.nwin * p.nb;
  const uint64_t mean = (items + buckets - 1) / buckets;
  uint64_t thr = 256;
  while (thr < 8 * mean) thr <<= 1;
  // Small problems (<= 2^16 buckets) cannot fill the chip, so the accumulation lasts as long as its longest
  // thread: send everything above three times the mean load to the cooperative kernels.  (n = 2^12, c = 12:
  // the partial top digit gives one bucket 4 x the mean, 180 points = a 3 ms chain, yet below the old
  // threshold; twice the mean would flood the cooperative path at c = 13, where 116 buckets carry 2.75 x.)
  if (buckets <= (1u << 16)) {
    thr = 3 * mean;
    if (thr < 48) thr = 48;
  }
  // short segments where the reduction is pure latency (one small MSM: a segment is a dependent chain of 2 x length
  // additions, and the tree sums behind it are cut into slices, msm_impl.hpp k_treesum: 2^14 buckets = 4 + 14 dependent
  // additions, against 16 + 23 with 8-bucket segments and one workgroup per job), 16 buckets once there are enough
  // segments to fill the chip
  p.seg_log = buckets <= (1u << 14) ? 1 : buckets <= (1u << 15) ? 2 : buckets <= (1u << 16) ? 3 : 4;
  // One windowed MSM by itself (the generic entry points up to 2^23 terms: 16 windows x 2^15 buckets) is a latency chain like a
  // small proof's: nothing else fills the chip while its reduction runs -- 32 dependent additions per 16-bucket segment
  // (0.52 ms at 2^20 terms) and 23 per tree-sum job (0.36 ms).  8-bucket segments and sliced job lists: 16 + 13.
  if (!p.shared && buckets <= (1u << 19) && p.seg_log > 3) p.seg_log = 3;
  // the partitioned big windows (13 x 2^19 buckets): 64-bucket segments -- the chains stay hidden behind 6 600 waves of them and
  // the tree sums walk a quarter of the segments (reduction of a 2^26-term MSM 5.66 -> 5.0 ms)
  if (!p.shared && p.c > 16) p.seg_log = 6;
  {
    const int seg_env = ZK_TUNE("ZKMI_SEG_LOG", 0);  // A/B library: segment length of big plans
    // (the segment arrays hold max(buckets / 16, 2^16) entries: msm_impl.hpp msm_max_segments)
    if (seg_env >= 1 && seg_env <= 7 && (buckets >> seg_env) <= (buckets / 16 > (1u << 16) ? buckets / 16 : (1u << 16))) p.seg_log = seg_env;
  }
  if ((1u << p.seg_log) > p.nb) p.seg_log = 0;
  p.heavy_thr = (uint32_t)thr;
  p.heavy_shift = 0;
  while ((thr >> p.heavy_shift) > 256) p.heavy_shift++;
}

MsmPlan msm_make_plan_c(uint64_t n, int c) {
  MsmPlan p;
  p.c = c;
  p.nwin = 255 / p.c + 1;
  p.nb = 1u << (p.c - 1);
  p.n = n;
  plan_set_heavy(p, n * (uint64_t)p.nwin);
  // (for every c in 17..22 the top window covers at most 15 bits: 0, 3, 8, 15, 3, 13)
  if (c > 16) p.top_spread_log = c - 1 - 15;
  return p;
}
MsmPlan msm_make_plan(uint64_t n) {
  // ZKMI_WINDOW_BITS (A/B library): tuning override (5..16, and 17..22 = the partitioned windows of run_windowed_big)
  const int v = ZK_TUNE("ZKMI_WINDOW_BITS", 0);
  const int forced = (v >= 5 && v <= MSM_MAX_WINDOW) ? v : 0;
  return msm_make_plan_c(n, forced ? forced : pick_window(n));
}

MsmPlan msm_make_plan_shared(uint64_t n) {
  // total buckets 2^(c-1) ~ n / 2: mean bucket load = ndigits, so a thread-per-bucket
  // accumulation has >= 4 full rounds of waves at n = 2^20 (c = 20: 13 digits instead of 16;
  // n = 2^22: c = 22, 12 digits)
  int lg = 0;
  while ((1ull << lg) < n) lg++;
  // Digit width c: all ndigits = floor(255 / c) + 1 digits of a scalar land in one set of 2^(c-1) buckets.
  // Cost model: (digits that can be non-zero) x n insertions + 2^(c-1) buckets x ~3.5 insertion-equivalents of
  // reduction work.  The LAST digit only covers top = 255 - (ndigits - 1) c bits of the (< 0.45 x 2^255) scalars:
  //   top = 0  : it is always zero (only the recoding bias lives there) -- a free digit;
  //   top small: every scalar's last digit falls into ~0.45 x 2^top buckets, each holding
  //              2^(c-1) / (ndigits x 0.45 x 2^top) times the mean load -- a few hundred oversized buckets that the
  //              thread-per-bucket kernel cannot take and the workgroup-per-bucket path handles badly (measured at
  //              c = 14: 19 digits, top = 3 -> groups of 2^14-constraint proofs 3x slower than 2^15 ones;
  //              c = 19, top = 8 -> 2^18 proofs 6x slower).  Such widths are skipped (ratio > 4).
  int c = 0;
  double best = 0;
  for (int cand = (lg > 9 ? (lg - 3 < 17 ? lg - 3 : 17) : 6); cand <= 22 && cand <= lg + 2; cand++) {
    if (cand < 6) continue;
    const int nd = 255 / cand + 1, top = 255 - (nd - 1) * cand;
    const double ratio = top == 0 ? 0.0 : (double)(1ull << (cand - 1)) / (nd * 0.45 * (double)(1ull << top));
    if (ratio > 4.0) continue;
    const double cost = (double)(top == 0 ? nd - 1 : nd) * (double)(n ? n : 1) + 3.5 * (double)(1ull << (cand - 1));
    if (c == 0 || cost < best) {
      c = cand;
      best = cost;
    }
  }
  if (c == 0) c = lg < 6 ? 6 : (lg > 22 ? 22 : lg);
  MsmPlan p;
  p.c = c;
  p.shared = true;
  p.ndigits = 255 / c + 1;
  const int nb_log = (c - 1 < 15) ? c - 1 : 15;
  p.nb = 1u << nb_log;
  p.nwin = 1 << (c - 1 - nb_log);
  p.n = n;
  plan_set_heavy(p, n * (uint64_t)p.ndigits);
  return p;
}

// plan of MsmSort::run_shared_batch: `batch` vectors of n scalars, vector v owns partitions [v * vec_parts, (v + 1) * vec_parts)
MsmPlan msm_make_plan_shared_batch(uint64_t n, uint32_t batch) {
  MsmPlan p = msm_make_plan_shared(n);
  p.vec_parts = p.nwin;
  p.nwin = (int)(batch * (uint32_t)p.vec_parts);
  plan_set_heavy(p, n * (uint64_t)p.ndigits * batch);
  return p;
}

// windowed plans with more than 2^15 buckets per window: partitions per window and in total (run_windowed_big)
static inline uint32_t big_parts_per_window(const MsmPlan& p) { return p.nb >> 15; }
static inline bool plan_is_big(const MsmPlan& p) { return !p.shared && p.c > 16; }
// 16-bit windows (2^15 buckets = one group per window) from 2^21 terms on take the same two-level record sort as the big plans
// (2^21: 0.87 -> 0.69 ms, 2^23: 4.1 -> 2.0; below that its fourteen short launches cost what they save -- 2^20: 0.45 ms either
// way over BLS12-381 scalars, 0.62 against 0.52 over BN254 ones.  A/B library: ZKMI_WIN_TWO_LEVEL = smallest log2(n), 30 = never)
static inline bool plan_two_level(const MsmPlan& p, uint64_t n) {
  return !p.shared && p.c == 16 && n >= (1ull << ZK_TUNE("ZKMI_WIN_TWO_LEVEL", 21)) && ZK_TUNE("ZKMI_BIG_SORT", 1) != 0;
}

static const uint64_t PLAN_STEPS[] = {1u << 8, 1u << 11, 1u << 14, 1u << 17, ~0ull};

uint64_t msm_max_buckets(uint64_t n) {
  uint64_t best = 0;
  for (uint64_t step : PLAN_STEPS) {
    uint64_t m = step < n ? step : n;
    MsmPlan p = msm_make_plan(m);
    uint64_t v = (uint64_t)p.nwin * p.nb;
    if (v > best) best = v;
    if (step >= n) break;
  }
  return best;
}

static uint64_t msm_max_entries(uint64_t n) {
  uint64_t best = 0;
  for (uint64_t step : PLAN_STEPS) {
    uint64_t m = step < n ? step : n;
    MsmPlan p = msm_make_plan(m);
    uint64_t v = (uint64_t)p.nwin * m;
    if (v > best) best = v;
    if (step >= n) break;
  }
  return best;
}

// more, smaller (chunk, window) tiles for big inputs: see shared_chunks below (A/B library: ZKMI_WIN_CHUNKS=m /
// ZKMI_REC_CHUNKS=m force the multiplier)
static uint32_t chunk_multiplier(uint64_t n, bool windowed) {
  const int v = windowed ? ZK_TUNE("ZKMI_WIN_CHUNKS", 0) : ZK_TUNE("ZKMI_REC_CHUNKS", 0);
  if (v >= 1 && v <= 64) return (uint32_t)v;
  if (n < (1ull << 22)) return 1u;
  if (windowed) return 4u;  // measured: 2^24 sort 8.9 -> 5.8 ms, 2^25 17.6 -> 12.3; no gain beyond 4x
  return n >= (1ull << 26) ? 16u : n >= (1ull << 25) ? 8u : 4u;
}

static uint32_t pick_chunks(const MsmPlan& p) {
  uint64_t nch = (uint64_t)((256 + p.nwin - 1) / p.nwin) * chunk_multiplier(p.n, true);  // ~1 tile per CU x multiplier
  const uint64_t max_by_n = (p.n + 1023) / 1024;
  if (nch > max_by_n) nch = max_by_n ? max_by_n : 1;
  return nch ? (uint32_t)nch : 1;
}

static uint32_t shared_chunks(uint32_t P, uint64_t n);
static uint64_t msm_max_hist(uint64_t n) {
  uint64_t best = 0;
  for (uint64_t step : PLAN_STEPS) {
    uint64_t m = step < n ? step : n;
    MsmPlan p = msm_make_plan(m);
    uint64_t v = (uint64_t)p.nwin * p.nb * (plan_is_big(p) ? shared_chunks((uint32_t)p.nwin * big_parts_per_window(p), m) : pick_chunks(p));
    if (v > best) best = v;
    if (step >= n) break;
  }
  // a forced window width (multi-GPU split) can pair the largest bucket count with any n
  MsmPlan q = msm_make_plan_c(n, 16);
  uint64_t v = (uint64_t)q.nwin * q.nb * pick_chunks(q);
  return v > best ? v : best;
}

void MsmSort::release() {
  if (count) (void)hipFree(count);
  if (perm) (void)hipFree(perm);
  if (heavy) (void)hipFree(heavy);
  if (part_total) (void)hipFree(part_total);
  if (order_bins) (void)hipFree(order_bins);
  if (blkcnt) (void)hipFree(blkcnt);
  if (fpart) (void)hipFree(fpart);
  if (rec_entry) (void)hipFree(rec_entry);
  if (rec_bkt) (void)hipFree(rec_bkt);
  if (rec_aux) (void)hipFree(rec_aux);
  if (big_ws) (void)hipFree(big_ws);
  big_ws = nullptr;
  if (begin) (void)hipFree(begin);
  if (blockhist) (void)hipFree(blockhist);
  if (sorted) (void)hipFree(sorted);
  count = begin = blockhist = sorted = perm = heavy = part_total = blkcnt = rec_entry = rec_bkt = rec_aux = order_bins = fpart = nullptr;
  cap_entries = cap_buckets = cap_hist = 0;
  has_shared = false;
}

// (chunk, partition) tiles of the single-vector shared sort: ~one tile per CU up to 2^22 terms, four times as many
// above.  The scatter pass writes 4-byte entries at random into a partition's slice of sorted[]; with more, smaller
// tiles the tiles in flight cover fewer partitions at a time and more of those partial writes meet in cache
// (2^24 terms: sort 10.0 -> 6.1 ms with 4x, 2^26: 39.4 -> 26.4 ms with 16x; at 2^20 the extra histogram traffic costs more than it saves: 0.63 -> 1.39 ms
// with 16x).  ZKMI_REC_CHUNKS=m forces the multiplier.
static uint32_t shared_chunks(uint32_t P, uint64_t n) {
  const uint32_t mult = chunk_multiplier(n, false);
  uint64_t nch = (uint64_t)((256 + P - 1) / P) * (P > 1 ? mult : 1);
  const uint64_t max_by_n = (n + 1023) / 1024;
  if (nch > max_by_n) nch = max_by_n ? max_by_n : 1;
  return (uint32_t)nch;
}

hipError_t MsmSort::reserve(uint64_t n, bool shared_too) {
  uint64_t ne = msm_max_entries(n), nbk = msm_max_buckets(n), nh = msm_max_hist(n);
  const uint64_t forced = (uint64_t)(255 / 16 + 1) * (1u << 15);
  if (nbk < forced) nbk = forced;  // allow plan_override = 16 for any n
  if (16 * n > ne) ne = 16 * n;
  // (the record buffers serve the shared-bucket sorts and the partitioned windows of big windowed plans alike)
  shared_too = shared_too || has_shared || plan_is_big(msm_make_plan(n)) || plan_two_level(msm_make_plan(n), n);
  if (shared_too) {
    // shared-bucket plan for the same n (only the prover uses it)
    const MsmPlan sp = msm_make_plan_shared(n);
    const uint64_t se = (uint64_t)sp.ndigits * n, sb = (uint64_t)sp.nwin * sp.nb;
    const uint64_t snch = shared_chunks((uint32_t)sp.nwin, n);
    if (se > ne) ne = se;
    if (sb > nbk) nbk = sb;
    if (sb * snch > nh) nh = sb * snch;
  }
  const bool need_records = shared_too && !has_shared;  // the record buffers only exist in shared mode
  if (!need_records && ne <= cap_entries && nbk <= cap_buckets && nh <= cap_hist) return hipSuccess;
  // Grow monotonically: a grouped key reserved room through reserve_batch() that the sizes derived from this n alone
  // do not cover, and that key is still alive (a 2^14 key with 64-proof groups owns 2^20 buckets; a later 2^21-term
  // MSM on the same context needs more entries but only 2^19 buckets).  No dimension ever shrinks.
  if (ne < cap_entries) ne = cap_entries;
  if (nbk < cap_buckets) nbk = cap_buckets;
  if (nh < cap_hist) nh = cap_hist;
  return allocate(ne, nbk, nh, shared_too);
}

// frees everything and allocates for (ne entries, nbk buckets, nh tile-histogram words); callers pass sizes that are
// at least the current capacities.  Nothing may be in flight on these buffers: every entry point that sorts also
// waits for its results before it returns, and hipFree waits for the device.
hipError_t MsmSort::allocate(uint64_t ne, uint64_t nbk, uint64_t nh, bool shared) {
  release();
  if (!ne) ne = 1;
  hipError_t e;
  if ((e = hipMalloc(&count, sizeof(uint32_t) * nbk)) != hipSuccess) return e;
  if ((e = hipMalloc(&begin, sizeof(uint32_t) * nbk)) != hipSuccess) return e;
  if ((e = hipMalloc(&perm, sizeof(uint32_t) * nbk)) != hipSuccess) return e;
  if ((e = hipMalloc(&part_total, sizeof(uint32_t) * PART_MAX)) != hipSuccess) return e;
  if ((e = hipMalloc(&order_bins, sizeof(uint32_t) * (MSM_HEAVY + 2) * (256 + 1))) != hipSuccess) return e;  // row 0: offsets, rows 1..: per-block key counts
  if ((e = hipMalloc(&blkcnt, sizeof(uint32_t) * FPART_BLOCKS * FINE_MAX_PARTS)) != hipSuccess) return e;  // also 256 x 64 of the coarse form
  static_assert(4 * FINE_BIG_PARTS >= 2 * FINE_MAX_PARTS, "fpart serves both fine-partition forms");
  if ((e = hipMalloc(&fpart, sizeof(uint32_t) * 4 * FINE_BIG_PARTS)) != hipSuccess) return e;  // bases, totals, cursors, oversize flags
  if (shared) {
    if ((e = hipMalloc(&rec_entry, sizeof(uint32_t) * ne)) != hipSuccess) return e;
    if ((e = hipMalloc(&rec_bkt, sizeof(uint32_t) * ne)) != hipSuccess) return e;
    // second record array of the two-level big windowed sort (its first level borrows sorted[] for the entries); only where
    // such a plan can occur: from 2^21 entries on (a 2^17-term slice of a split 2^24-term MSM)
    if (ne >= (1ull << 21) && (e = hipMalloc(&rec_aux, sizeof(uint32_t) * ne)) != hipSuccess) return e;
    if (rec_aux && (e = hipMalloc(&big_ws, sizeof(uint32_t) * (1 + BIG_MAX + (size_t)BIG_MAX * BIG_NCH * FINE_NB))) != hipSuccess) return e;
  }
  has_shared = shared;
  if ((e = hipMalloc(&heavy, sizeof(uint32_t) * (nbk + 1))) != hipSuccess) return e;  // [0] = list length
  if ((e = hipMalloc(&blockhist, sizeof(uint32_t) * nh)) != hipSuccess) return e;
  if ((e = hipMalloc(&sorted, sizeof(uint32_t) * ne)) != hipSuccess) return e;
  cap_entries = ne;
  cap_buckets = nbk;
  cap_hist = nh;
  return hipSuccess;
}

// 160 KiB dynamic-LDS opt-in of the histogram kernels for the CURRENT device (function attributes are
// per device: zkmi_ctx_create calls this after hipSetDevice, next to ntt_enable_big_lds)
hipError_t msm_sort_enable_big_lds() {
  const void* fns[] = {reinterpret_cast<const void*>(k_bucket_pass<false>),
                       reinterpret_cast<const void*>(k_bucket_pass<true>),
                       reinterpret_cast<const void*>(k_bucket_pass_shared<false>),
                       reinterpret_cast<const void*>(k_bucket_pass_shared<true>),
                       reinterpret_cast<const void*>(k_bucket_pass_rec<false>),
                       reinterpret_cast<const void*>(k_bucket_pass_rec<true>)};
  for (const void* f : fns) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  // the staged kernels keep static LDS beside their stage: ask for exactly the stage
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fpart_write_staged<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(uint32_t) * 2 * 1024 * FPASS_MAXD));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fpart_write_staged<2048>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(uint32_t) * 2 * 1024 * 13));
    if (e != hipSuccess) return e;
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_pass<false, true, true, uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(uint32_t) * FINE_BIG_PARTS));
    if (e != hipSuccess) return e;
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_write_staged<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(uint32_t) * 2 * 1024 * WSTAGE_MAXD));
    if (e != hipSuccess) return e;
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rec_split<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(uint32_t) * 2 * SPLIT_B));
    if (e != hipSuccess) return e;
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fpart_sort<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(uint32_t) * FINE_STAGE));
    if (e != hipSuccess) return e;
  }
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_fpart_sort<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)(sizeof(uint32_t) * FINE_STAGE));
}

hipError_t MsmSort::wait_readers(hipStream_t st) {
  for (hipEvent_t ev : readers) {
    const hipError_t e = hipStreamWaitEvent(st, ev, 0);
    if (e != hipSuccess) return e;
  }
  readers.clear();
  return hipSuccess;
}

hipError_t MsmSort::run(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof) {
  {
    const hipError_t er = wait_readers(st);
    if (er != hipSuccess) return er;
  }
  plan = plan_override ? msm_make_plan_c(n, plan_override) : msm_make_plan(n);
  if (win_count > 0) {
    if (win_first < 0 || win_first + win_count > plan.nwin) return hipErrorInvalidValue;
    plan.nwin_total = plan.nwin;
    plan.win_first = win_first;
    plan.nwin = win_count;
  }
  if (plan_is_big(plan) || (plan_two_level(plan, n) && rec_aux != nullptr && (uint64_t)plan.nwin * n <= cap_entries))
    return run_windowed_big(d_scalars, n, st, prof);
  const uint32_t nb = plan.nb, nwin = (uint32_t)plan.nwin;
  const uint32_t tot_b = nwin * nb;
  const uint32_t nch = pick_chunks(plan);
  const uint32_t chunk = (uint32_t)((n + nch - 1) / nch);
  if ((uint64_t)nwin * n > cap_entries || tot_b > cap_buckets || (uint64_t)tot_b * nch > cap_hist) return hipErrorInvalidValue;
  // M = sum_w (2^(c-1) - 1) 2^(c w), 9 limbs
  RecodeConst rc;
  for (int j = 0; j < 9; j++) rc.m[j] = 0;
  for (uint32_t w = 0; w < (uint32_t)plan.total_windows(); w++) {  // (the recoding bias covers ALL digit positions)
    const uint64_t v = (1ull << (plan.c - 1)) - 1;
    const int bit = (int)w * plan.c, limb = bit >> 5, sh = bit & 31;
    if (limb < 9) {
      const uint64_t lo = v << sh;  // c <= 16, sh <= 31: fits 64 bits
      uint64_t carry = lo;
      for (int j = limb; j < 9 && carry; j++) {
        const uint64_t sum = (uint64_t)rc.m[j] + (uint32_t)carry;
        rc.m[j] = (uint32_t)sum;
        carry = (carry >> 32) + (sum >> 32);
      }
    }
  }
  if (prof) prof->begin(PH_MSM_SORT, st);
  const size_t lds = sizeof(uint32_t) * nb;
  const dim3 grid(nch, nwin);
  hipLaunchKernelGGL(k_bucket_pass<false>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c, nb, chunk, rc,
                     blockhist, sorted, (uint32_t)plan.win_first);
  hipLaunchKernelGGL(k_bucket_totals, dim3((tot_b + 255) / 256), dim3(256), 0, st, blockhist, count, nb, nch, tot_b);
  hipLaunchKernelGGL(k_window_scan, dim3(nwin), dim3(1024), 0, st, count, begin, nb, (uint32_t)n,
                     (const uint32_t*)nullptr);
  const uint32_t tot_h = tot_b * nch;
  hipLaunchKernelGGL(k_bucket_bases, dim3((tot_h + 255) / 256), dim3(256), 0, st, blockhist, begin, nb, nch, tot_h);
  hipError_t e0 = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
  if (e0 != hipSuccess) return e0;
  hipLaunchKernelGGL(k_bucket_pass<true>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c, nb, chunk, rc,
                     blockhist, sorted, (uint32_t)plan.win_first);
  if (prof) prof->end(PH_MSM_SORT, st);
  return hipGetLastError();
}

// Windowed plan with c > 16 (`plan` is set by run()): window w owns 2^(c-1) buckets = ppw partitions of 2^15, the
// LDS-histogram tiles work on (chunk, window x partition).  The digits of ALL windows are extracted in one record
// pre-pass (two reads of the scalars in total, where the plain windowed sort reads them twice per window), the records
// of a (window, partition) group are then counted and scattered like the partitions of the shared-bucket record sort.
// sorted[] is packed group by group; bucket ids are window-major (w * 2^(c-1) + |digit| - 1), which is what the
// reductions of the windowed schedule expect.
hipError_t MsmSort::run_windowed_big(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof) {
  // (tiles of 2^14 buckets -- big_nb_log -- only where the top window, whose 15-bit digits are spread over the partitions by
  // point index, is not among this sort's windows)
  const bool has_top = plan.win_first + plan.nwin == plan.total_windows();
  const int nb_log = (big_nb_log == 14 && !has_top) ? 14 : 15;
  const uint32_t nb = 1u << nb_log, nwin = (uint32_t)plan.nwin;
  const uint32_t P = nwin * (plan.nb >> nb_log);
  const uint32_t tot_b = nwin * plan.nb;
  const uint32_t nch = shared_chunks(P, n);
  if (P > PART_MAX || (uint64_t)nwin * n > cap_entries || tot_b > cap_buckets || !rec_entry) return hipErrorInvalidValue;
  RecodeConst rc;
  for (int j = 0; j < 9; j++) rc.m[j] = 0;
  for (uint32_t w = 0; w < (uint32_t)plan.total_windows(); w++) {  // (the recoding bias covers ALL digit positions)
    const uint64_t v = (1ull << (plan.c - 1)) - 1;
    const int bit = (int)w * plan.c, limb = bit >> 5, sh = bit & 31;
    if (limb < 9) {
      uint64_t carry = v << sh;  // c <= 22, sh <= 31: fits 64 bits
      for (int j = limb; j < 9 && carry; j++) {
        const uint64_t sum = (uint64_t)rc.m[j] + (uint32_t)carry;
        rc.m[j] = (uint32_t)sum;
        carry = (carry >> 32) + (sum >> 32);
      }
    }
  }
  if (prof) prof->begin(PH_MSM_SORT, st);
  uint32_t nblk = (uint32_t)((n + 4095) / 4096);
  if (nblk > 256) nblk = 256;
  if (!nblk) nblk = 1;
  const uint32_t chunk = (uint32_t)((n + nblk - 1) / nblk);
  const int w_top_pos = plan.total_windows() - 1;
  // (the counting pass needs 42 registers per lane: beside an occupancy-capped accumulation -- nb_log 14 -- only two of its waves
  // fit a SIMD's free registers, so its workgroups are 512 threads there)
  // Fine-partition form (see k_rec_split): groups of 2^15 buckets -> their 32 fine partitions -> k_fpart_sort.  The records of the
  // first level go to (sorted, rec_aux), the second level's to (rec_entry, rec_bkt), the entries end in sorted[].
  // buckets per fine partition: 2^10, or fewer (down to 2^8) while a fine partition would hold more than 32 768 records on average
  int fine_log = FINE_LOG;
  while (fine_log > 7 && (n >> (plan.c - 1 - fine_log)) > 32768) fine_log--;
  {
    const int fl = ZK_TUNE("ZKMI_BIG_FINE_LOG", 0);
    if (fl >= 7 && fl <= FINE_LOG) fine_log = fl;
  }
  const uint32_t NPF = tot_b >> fine_log;
  if (nb_log == 15 && NPF <= FINE_BIG_PARTS && rec_aux != nullptr && big_ws != nullptr && ZK_TUNE("ZKMI_BIG_SORT", 1) != 0) {
    const int fan_log = nb_log - fine_log;
    hipError_t e = hipMemsetAsync(fpart + NPF, 0, sizeof(uint32_t) * NPF, st);
    if (e != hipSuccess) return e;
    // (bucket ids travel as 16-bit words through the three levels; rec_aux / rec_bkt are sized for 32-bit ones)
    uint16_t* const key_a = reinterpret_cast<uint16_t*>(rec_aux);
    uint16_t* const key_b = reinterpret_cast<uint16_t*>(rec_bkt);
    hipLaunchKernelGGL((k_part_pass<false, true, true, uint16_t>), dim3(nblk), dim3(1024), sizeof(uint32_t) * NPF, st, d_scalars, (uint32_t)n, plan.c, (int)nwin, nb,
                       nb_log, P, chunk, rc, blkcnt, sorted, key_a, plan.win_first, w_top_pos, fpart + NPF, fine_log);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(PART_MAX), 0, st, blkcnt, nblk, P, part_total);
    hipLaunchKernelGGL(k_fpart_scan_base, dim3(1), dim3(1024), 0, st, NPF, fpart, (uint32_t*)nullptr, 0u, big_ws);
    hipLaunchKernelGGL(k_fine_cursors, dim3((NPF + 255) / 256), dim3(256), 0, st, fpart, NPF, big_ws, BIG_MAX);
    if (nwin <= WSTAGE_MAXD && ZK_TUNE("ZKMI_BIG_WSTAGE", 1) != 0)
      hipLaunchKernelGGL(k_part_write_staged<uint16_t>, dim3(nblk), dim3(1024), sizeof(uint32_t) * 2 * 1024 * nwin, st, d_scalars, (uint32_t)n, plan.c, (int)nwin, nb,
                         nb_log, P, chunk, rc, (const uint32_t*)blkcnt, sorted, key_a, plan.win_first, w_top_pos);
    else
      hipLaunchKernelGGL((k_part_pass<true, true, false, uint16_t>), dim3(nblk), dim3(1024), 0, st, d_scalars, (uint32_t)n, plan.c, (int)nwin, nb, nb_log, P,
                         chunk, rc, blkcnt, sorted, key_a, plan.win_first, w_top_pos);
    // batches per group on average -> workgroups per group (at most 16: 3 328 workgroups at 13 windows)
    uint64_t per_group = ((uint64_t)nwin * n / P + SPLIT_B - 1) / SPLIT_B;
    const uint32_t nch2 = per_group > 16 ? 16u : (per_group ? (uint32_t)per_group : 1u);
    hipLaunchKernelGGL(k_rec_split<uint16_t>, dim3(nch2, P), dim3(1024), sizeof(uint32_t) * 2 * SPLIT_B, st, (const uint32_t*)sorted, (const uint16_t*)key_a,
                       (const uint32_t*)part_total, fpart + 2 * NPF, rec_entry, key_b, fan_log, fine_log, ZK_TUNE("ZKMI_SPLIT_PLAIN_RANK", 0));
    // (oversized partitions first: see k_fine_cursors; big_ws = the list, then BIG_MAX x BIG_NCH x 1024 chunk counters)
    uint32_t* const bighist = big_ws + 1 + BIG_MAX;
    hipLaunchKernelGGL(k_big_hist<uint16_t>, dim3(BIG_NCH, BIG_MAX), dim3(1024), 0, st, (const uint16_t*)key_b, (const uint32_t*)fpart, NPF, (const uint32_t*)big_ws, bighist);
    hipLaunchKernelGGL(k_big_scan, dim3(BIG_MAX), dim3(1024), 0, st, (const uint32_t*)fpart, NPF, (const uint32_t*)big_ws, bighist, count, begin, fine_log);
    hipLaunchKernelGGL(k_big_scatter<uint16_t>, dim3(BIG_NCH, BIG_MAX), dim3(1024), 0, st, (const uint32_t*)rec_entry, (const uint16_t*)key_b, (const uint32_t*)fpart, NPF,
                       (const uint32_t*)big_ws, (const uint32_t*)bighist, sorted);
    hipLaunchKernelGGL(k_fpart_sort<uint16_t>, dim3(NPF), dim3(1024), sizeof(uint32_t) * FINE_STAGE, st, (const uint32_t*)rec_entry, (const uint16_t*)key_b,
                       (const uint32_t*)fpart, NPF, count, begin, sorted, 1, fine_log, (const uint32_t*)(fpart + 3 * NPF));
    e = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
    if (e != hipSuccess) return e;
    if (prof) prof->end(PH_MSM_SORT, st);
    return hipGetLastError();
  }
  if ((uint64_t)tot_b * nch > cap_hist) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_part_pass<false, true>), dim3(nblk), dim3(nb_log == 14 ? 512 : 1024), 0, st, d_scalars, (uint32_t)n, plan.c, (int)nwin, nb, nb_log, P,
                     chunk, rc, blkcnt, rec_entry, rec_bkt, plan.win_first, w_top_pos);
  hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(PART_MAX), 0, st, blkcnt, nblk, P, part_total);
  hipLaunchKernelGGL((k_part_pass<true, true>), dim3(nblk), dim3(1024), 0, st, d_scalars, (uint32_t)n, plan.c, (int)nwin, nb, nb_log, P,
                     chunk, rc, blkcnt, rec_entry, rec_bkt, plan.win_first, w_top_pos);
  const size_t lds = sizeof(uint32_t) * nb;
  const dim3 grid(nch, P);
  hipLaunchKernelGGL(k_bucket_pass_rec<false>, grid, dim3(1024), lds, st, rec_entry, rec_bkt, part_total, nb, blockhist, sorted);
  hipLaunchKernelGGL(k_bucket_totals, dim3((tot_b + 255) / 256), dim3(256), 0, st, blockhist, count, nb, nch, tot_b);
  hipLaunchKernelGGL(k_window_scan, dim3(P), dim3(1024), 0, st, count, begin, nb, (uint32_t)n, (const uint32_t*)part_total);
  const uint64_t tot_h = (uint64_t)tot_b * nch;
  hipLaunchKernelGGL(k_bucket_bases, dim3((unsigned)((tot_h + 255) / 256)), dim3(256), 0, st, blockhist, begin, nb, nch, (uint32_t)tot_h);
  hipError_t e0 = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
  if (e0 != hipSuccess) return e0;
  hipLaunchKernelGGL(k_bucket_pass_rec<true>, grid, dim3(1024), lds, st, rec_entry, rec_bkt, part_total, nb, blockhist, sorted);
  if (prof) prof->end(PH_MSM_SORT, st);
  return hipGetLastError();
}

hipError_t MsmSort::run_shared(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof) {
  {
    const hipError_t er = wait_readers(st);
    if (er != hipSuccess) return er;
  }
  plan = msm_make_plan_shared(n);
  const uint32_t nb = plan.nb, P = (uint32_t)plan.nwin;
  int nb_log = 0;
  while ((1u << nb_log) < nb) nb_log++;
  const uint32_t tot_b = P * nb;
  const uint32_t nch = shared_chunks(P, n);
  const uint32_t chunk = (uint32_t)((n + nch - 1) / nch);
  if ((uint64_t)plan.ndigits * n > cap_entries || tot_b > cap_buckets || (uint64_t)tot_b * nch > cap_hist || (P > 1 && !rec_entry))
    return hipErrorInvalidValue;
  RecodeConst rc;
  for (int j = 0; j < 9; j++) rc.m[j] = 0;
  for (int w = 0; w < plan.ndigits; w++) {
    const uint64_t v = (1ull << (plan.c - 1)) - 1;
    const int bit = w * plan.c, limb = bit >> 5, sh = bit & 31;
    if (limb < 9) {
      uint64_t carry = v << sh;  // c <= 20, sh <= 31: fits 64 bits
      for (int j = limb; j < 9 && carry; j++) {
        const uint64_t sum = (uint64_t)rc.m[j] + (uint32_t)carry;
        rc.m[j] = (uint32_t)sum;
        carry = (carry >> 32) + (sum >> 32);
      }
    }
  }
  if (prof) prof->begin(PH_MSM_SORT, st);
  const size_t lds = sizeof(uint32_t) * nb;
  const dim3 grid(nch, P);
  const bool records = P > 1;
  // The fine-partition sort (kernels above) serves the plans whose fine partitions hold at most two stage rounds of
  // records: the prover up to N = 2^21 and prepared MSMs of the same sizes.  Alone: 0.60 -> 0.28 ms at 2^20 terms, 1.24 ->
  // 0.87 ms at 2^21; inside the proof pipeline +4.8 % proofs/s at N = 2^20.  Beyond that the digit width stays at 20 bits
  // (2^19 buckets) while the records grow with n, every extra stage round re-reads the partition's records, and the record
  // sort wins again (2^22: 3.3 vs 2.0 ms, 2^24: 42 vs 6 ms).  ZKMI_SORT_FINE=0 / =2: never / whenever it is applicable.
  const int fine_mode = ZK_TUNE("ZKMI_SORT_FINE", 1);
  const uint32_t NP = tot_b >> FINE_LOG;
  const bool fine_fits = NP <= FINE_MAX_PARTS && (uint64_t)plan.ndigits * n / (NP ? NP : 1) <= 2 * FINE_ROUND &&
                         plan.ndigits <= (int)FPASS_MAXD;
  if (records && fine_mode != 0 && (fine_mode == 2 ? NP <= FINE_MAX_PARTS : fine_fits) && fpart != nullptr) {
    uint32_t nblk = (uint32_t)((n + 4095) / 4096);
    if (nblk > FPART_BLOCKS) nblk = FPART_BLOCKS;
    const uint32_t chunk_a = (uint32_t)((n + nblk - 1) / nblk);
    const size_t lds_np = sizeof(uint32_t) * NP;
    hipLaunchKernelGGL(k_fpart_pass<false>, dim3(nblk), dim3(1024), lds_np, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, NP, chunk_a,
                       rc, blkcnt, (const uint32_t*)fpart, rec_entry, rec_bkt);
    hipLaunchKernelGGL(k_fpart_scan_rows, dim3(NP), dim3(64), 0, st, blkcnt, nblk, NP, fpart);
    // Oversized fine partitions (a witness of bits: 630 000 records in one bucket) stay with the round form of k_fpart_sort HERE:
    // the k_big_* kernels (A/B library: ZKMI_SORT_BIG=1, a grid of BIG_PROVER x 64 workgroups) are four more launches per sort in
    // the proof pipeline, where an empty launch still has to be placed -- measured 57.3 / 57.6 against 57.9 / 58.0 proofs/s at 2^20
    // for 174 / 167 against 163 / 166 on the bits relation (profiles/r05/experiments/prover_sort_big_partitions_ab.txt).
    constexpr uint32_t BIG_PROVER = 4;
    const bool big_on = big_ws != nullptr && ZK_TUNE("ZKMI_SORT_BIG", 0) != 0;
    hipLaunchKernelGGL(k_fpart_scan_base, dim3(1), dim3(1024), 0, st, NP, fpart, part_total, P, big_on ? big_ws : (uint32_t*)nullptr);
    const bool stage_on = ZK_TUNE("ZKMI_SORT_STAGE", 1) != 0;
    const size_t stage_bytes = sizeof(uint32_t) * 2 * 1024 * (size_t)plan.ndigits;
    if (stage_on && NP <= 512 && plan.ndigits <= (int)FPASS_MAXD)
      hipLaunchKernelGGL(k_fpart_write_staged<512>, dim3(nblk), dim3(1024), stage_bytes, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, NP,
                         chunk_a, rc, (const uint32_t*)blkcnt, (const uint32_t*)fpart, rec_entry, rec_bkt);
    else if (stage_on && NP <= 2048 && plan.ndigits <= 13)  // 32 KB of cursors + <= 104 KB of stage
      hipLaunchKernelGGL(k_fpart_write_staged<2048>, dim3(nblk), dim3(1024), stage_bytes, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, NP,
                         chunk_a, rc, (const uint32_t*)blkcnt, (const uint32_t*)fpart, rec_entry, rec_bkt);
    else
      hipLaunchKernelGGL(k_fpart_pass<true>, dim3(nblk), dim3(1024), lds_np, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, NP, chunk_a,
                         rc, blkcnt, (const uint32_t*)fpart, rec_entry, rec_bkt);
    if (big_on) {
      uint32_t* const bighist = big_ws + 1 + BIG_MAX;
      hipLaunchKernelGGL(k_fine_cursors, dim3((NP + 255) / 256), dim3(256), 0, st, fpart, NP, big_ws, BIG_PROVER);
      hipLaunchKernelGGL(k_big_hist<uint32_t>, dim3(BIG_NCH, BIG_PROVER), dim3(1024), 0, st, (const uint32_t*)rec_bkt, (const uint32_t*)fpart, NP, (const uint32_t*)big_ws,
                         bighist);
      hipLaunchKernelGGL(k_big_scan, dim3(BIG_PROVER), dim3(1024), 0, st, (const uint32_t*)fpart, NP, (const uint32_t*)big_ws, bighist, count, begin, FINE_LOG);
      hipLaunchKernelGGL(k_big_scatter<uint32_t>, dim3(BIG_NCH, BIG_PROVER), dim3(1024), 0, st, (const uint32_t*)rec_entry, (const uint32_t*)rec_bkt, (const uint32_t*)fpart,
                         NP, (const uint32_t*)big_ws, (const uint32_t*)bighist, sorted);
    }
    hipLaunchKernelGGL(k_fpart_sort<uint32_t>, dim3(NP), dim3(1024), stage_on ? sizeof(uint32_t) * FINE_STAGE : 0, st, (const uint32_t*)rec_entry,
                       (const uint32_t*)rec_bkt, (const uint32_t*)fpart, NP, count, begin, sorted, stage_on ? 1 : 0, FINE_LOG,
                       big_on ? (const uint32_t*)(fpart + 3 * NP) : (const uint32_t*)nullptr);
    hipError_t e1 = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
    if (e1 != hipSuccess) return e1;
    if (prof) prof->end(PH_MSM_SORT, st);
    return hipGetLastError();
  }
  uint32_t nblk_a = 0, chunk_a = 0;
  if (records) {
    nblk_a = (uint32_t)((n + 4095) / 4096);
    if (nblk_a > 256) nblk_a = 256;
    chunk_a = (uint32_t)((n + nblk_a - 1) / nblk_a);
    hipLaunchKernelGGL(k_part_pass<false>, dim3(nblk_a), dim3(1024), 0, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits,
                       nb, nb_log, P, chunk_a, rc, blkcnt, rec_entry, rec_bkt);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(PART_MAX), 0, st, blkcnt, nblk_a, P, part_total);
    hipLaunchKernelGGL(k_part_pass<true>, dim3(nblk_a), dim3(1024), 0, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits,
                       nb, nb_log, P, chunk_a, rc, blkcnt, rec_entry, rec_bkt);
    hipLaunchKernelGGL(k_bucket_pass_rec<false>, grid, dim3(1024), lds, st, rec_entry, rec_bkt, part_total, nb, blockhist,
                       sorted);
  } else {
    hipLaunchKernelGGL(k_bucket_pass_shared<false>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c,
                       plan.ndigits, nb, nb_log, chunk, rc, blockhist, sorted, (uint64_t)0, P);
  }
  hipLaunchKernelGGL(k_bucket_totals, dim3((tot_b + 255) / 256), dim3(256), 0, st, blockhist, count, nb, nch, tot_b);
  if (!records) hipLaunchKernelGGL(k_part_totals, dim3(P), dim3(1024), 0, st, count, part_total, nb);
  hipLaunchKernelGGL(k_window_scan, dim3(P), dim3(1024), 0, st, count, begin, nb, (uint32_t)n,
                     (const uint32_t*)part_total);
  const uint32_t tot_h = tot_b * nch;
  hipLaunchKernelGGL(k_bucket_bases, dim3((tot_h + 255) / 256), dim3(256), 0, st, blockhist, begin, nb, nch, tot_h);
  hipError_t e0 = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
  if (e0 != hipSuccess) return e0;
  if (records) {
    hipLaunchKernelGGL(k_bucket_pass_rec<true>, grid, dim3(1024), lds, st, rec_entry, rec_bkt, part_total, nb, blockhist,
                       sorted);
  } else {
    hipLaunchKernelGGL(k_bucket_pass_shared<true>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c,
                       plan.ndigits, nb, nb_log, chunk, rc, blockhist, sorted, (uint64_t)0, P);
  }
  if (prof) prof->end(PH_MSM_SORT, st);
  return hipGetLastError();
}

// Shared-bucket sort of `batch` independent scalar vectors of n elements each (vector b starts at
// d_scalars + b * stride_words) over the SAME bases: one bucket set per vector, laid out as the partitions of one
// combined plan (plan.nwin = batch), so that a single accumulation / reduction launch serves the whole batch.
// A vector whose own plan has several partitions (2^15-bucket ranges, n > 2^16) keeps them: partition q of the
// combined plan = range q % vec_parts of vector q / vec_parts.  Buffers must have been reserved for it.
hipError_t MsmSort::run_shared_batch(const uint32_t* d_scalars, uint64_t n, uint64_t stride_words, uint32_t batch, hipStream_t st,
                                     PhaseTimer* prof) {
  {
    const hipError_t er = wait_readers(st);
    if (er != hipSuccess) return er;
  }
  if (batch == 0 || (uint64_t)batch * msm_make_plan_shared(n).nwin > 64) return hipErrorInvalidValue;
  plan = msm_make_plan_shared_batch(n, batch);
  const uint32_t vparts = (uint32_t)plan.vec_parts;
  const uint32_t nb = plan.nb, P = batch * vparts;
  int nb_log = 0;
  while ((1u << nb_log) < nb) nb_log++;
  const uint32_t tot_b = P * nb;
  uint32_t nch = (512 + P - 1) / P;  // ~2 tiles per CU over the whole batch
  const uint64_t max_by_n = (n + 1023) / 1024;
  if (nch > max_by_n) nch = (uint32_t)(max_by_n ? max_by_n : 1);
  const uint32_t chunk = (uint32_t)((n + nch - 1) / nch);
  if ((uint64_t)plan.ndigits * n * batch > cap_entries || tot_b > cap_buckets || (uint64_t)tot_b * nch > cap_hist)
    return hipErrorInvalidValue;
  RecodeConst rc;
  for (int j = 0; j < 9; j++) rc.m[j] = 0;
  for (int w = 0; w < plan.ndigits; w++) {
    const uint64_t v = (1ull << (plan.c - 1)) - 1;
    const int bit = w * plan.c, limb = bit >> 5, sh = bit & 31;
    if (limb < 9) {
      uint64_t carry = v << sh;
      for (int j = limb; j < 9 && carry; j++) {
        const uint64_t sum = (uint64_t)rc.m[j] + (uint32_t)carry;
        rc.m[j] = (uint32_t)sum;
        carry = (carry >> 32) + (sum >> 32);
      }
    }
  }
  const bool dbg = debug_level() >= 1;
  auto chk = [&](const char* what) {
    if (!dbg) return;
    const hipError_t le = hipGetLastError();
    fprintf(stderr, "[zkmi] run_shared_batch %s: %s (n=%llu batch=%u nb=%u nch=%u chunk=%u c=%d nd=%d)\n", what, hipGetErrorString(le),
            (unsigned long long)n, batch, nb, nch, chunk, plan.c, plan.ndigits);
  };
  chk("entry");
  if (prof) prof->begin(PH_MSM_SORT, st);
  const size_t lds = sizeof(uint32_t) * nb;
  const dim3 grid(nch, P);
  hipLaunchKernelGGL(k_bucket_pass_shared<false>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, nb,
                     nb_log, chunk, rc, blockhist, sorted, stride_words, vparts);
  chk("pass1");
  hipLaunchKernelGGL(k_bucket_totals, dim3((tot_b + 255) / 256), dim3(256), 0, st, blockhist, count, nb, nch, tot_b);
  hipLaunchKernelGGL(k_part_totals, dim3(P), dim3(1024), 0, st, count, part_total, nb);
  hipLaunchKernelGGL(k_window_scan, dim3(P), dim3(1024), 0, st, count, begin, nb, (uint32_t)n, (const uint32_t*)part_total);
  const uint32_t tot_h = tot_b * nch;
  hipLaunchKernelGGL(k_bucket_bases, dim3((tot_h + 255) / 256), dim3(256), 0, st, blockhist, begin, nb, nch, tot_h);
  chk("scans");
  hipError_t e0 = bucket_order(count, perm, heavy, order_bins, tot_b, plan.heavy_thr, plan.heavy_shift, st);
  if (e0 != hipSuccess) return e0;
  hipLaunchKernelGGL(k_bucket_pass_shared<true>, grid, dim3(1024), lds, st, d_scalars, (uint32_t)n, plan.c, plan.ndigits, nb,
                     nb_log, chunk, rc, blockhist, sorted, stride_words, vparts);
  if (prof) prof->end(PH_MSM_SORT, st);
  return hipGetLastError();
}

// room for run_shared_batch(n, batch) on top of what reserve() provides
hipError_t MsmSort::reserve_batch(uint64_t n, uint32_t batch) {
  const MsmPlan sp = msm_make_plan_shared(n);
  const uint64_t pmax = (uint64_t)batch * sp.nwin;  // partitions of a full group
  if (pmax > 64) return hipErrorInvalidValue;
  const uint64_t ne = (uint64_t)sp.ndigits * n * batch, nbk = (uint64_t)sp.nb * pmax;
  // tiles = (partitions P) x (chunks per vector = min(ceil(512 / P), ceil(n / 1024))) for any P <= pmax:
  // P x ceil(512 / P) <= 512 + P
  const uint64_t mx = (n + 1023) / 1024 ? (n + 1023) / 1024 : 1;
  const uint64_t tiles = (512 + pmax) < pmax * mx ? (512 + pmax) : pmax * mx;
  const uint64_t nh = (uint64_t)sp.nb * tiles;
  if (ne <= cap_entries && nbk <= cap_buckets && nh <= cap_hist) return hipSuccess;
  // grow, never shrink (the buffers also serve other keys of this context)
  return allocate(ne > cap_entries ? ne : cap_entries, nbk > cap_buckets ? nbk : cap_buckets, nh > cap_hist ? nh : cap_hist,
                  has_shared);
}

// ---------------------------------------------------------------------------
PhaseTimer::~PhaseTimer() {
  for (int i = 0; i < n_created; i++) {
    (void)hipEventDestroy(ev0[i]);
    (void)hipEventDestroy(ev1[i]);
  }
}
void PhaseTimer::begin(int phase, hipStream_t st) {
  if (!enabled || n_pending >= MAX_PENDING) return;
  if (n_pending >= n_created) {
    (void)hipEventCreate(&ev0[n_created]);
    (void)hipEventCreate(&ev1[n_created]);
    n_created++;
  }
  phase_of[n_pending] = phase;
  (void)hipEventRecord(ev0[n_pending], st);
}
void PhaseTimer::end(int phase, hipStream_t st) {
  if (!enabled || n_pending >= MAX_PENDING) return;
  (void)phase;
  (void)hipEventRecord(ev1[n_pending], st);
  n_pending++;
}
void PhaseTimer::collect() {
  for (int i = 0; i < n_pending; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev0[i], ev1[i]) == hipSuccess) {
      total_ms[phase_of[i]] += ms;
      count[phase_of[i]]++;
    }
  }
  n_pending = 0;
}
void PhaseTimer::reset() {
  n_pending = 0;
  for (int i = 0; i < 16; i++) {
    total_ms[i] = 0;
    count[i] = 0;
  }
}

}  // namespace zkmi

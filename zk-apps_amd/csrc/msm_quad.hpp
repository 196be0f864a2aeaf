// zkmi — the reduction-side kernels of the MSM with every complete addition split over the lanes of a quad (G1, BN254 G1) or
// an octet (G2) -- quad.hpp: segment sums, tree sums, the redo pass and the summing of heavy-bucket partials.  Included by
// msm_impl.hpp.
//
// All of them are ONE-WAVE workgroups (16 quads / 8 octets) without LDS -- the shape of the accumulation kernels; the G1
// ones at most 168 registers -- so they are placed in the slot a retiring accumulation wave frees instead of waiting for the
// accumulation grids to drain (the one-lane forms: 230-320 registers, 128/256-thread workgroups), and a
// dependent addition is a chain of 4 field products instead of 14.
#pragma once
#include "quad.hpp"

namespace zkmi {

// PT = XYZZQ<F, 0, EXT2>: a point takes PT::LANES lanes (4: G1 / BN254 G1 quads; 8: G2 octets), a wave holds PT::PER_WAVE points
template <class PT>
using HostPointOf = XYZZ<typename HostFieldOf<typename PT::Elem>::type>;

// point t handles buckets [t*seg, (t+1)*seg); buckets2 / buckets3: see k_segreduce
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_segreduce_q(const typename PT::Point* __restrict__ buckets, typename PT::Point* __restrict__ segsum, typename PT::Point* __restrict__ segw,
              uint32_t total_segs, int seg, const typename PT::Point* __restrict__ buckets2, const typename PT::Point* __restrict__ buckets3) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / PT::LANES;
  if (t >= total_segs) return;  // uniform per point
  PT run = PT::infinity();
  PT acc = PT::infinity();
  for (int i = seg - 1; i >= 0; i--) {
    run.add(PT::load(buckets + (size_t)t * seg + i));
    if (buckets2) run.add(PT::load(buckets2 + (size_t)t * seg + i));
    if (buckets3) run.add(PT::load(buckets3 + (size_t)t * seg + i));
    acc.add(run);
  }
  run.store(segsum + t);
  acc.store(segw + t);
}

// the lanes of point 0 convert their values to the host representation and write them to the device copy and the pinned host slot
template <class PT>
__device__ __forceinline__ void quad_store_host(const PT& acc, HostPointOf<PT>* dev, HostPointOf<PT>* host) {
  const auto o = fq_from_fq28(acc.v);  // one base-field value in the host's 32-bit-limb form
  using HB = decltype(fq_from_fq28(acc.v));
  static_assert(sizeof(HB) % 16 == 0, "16-byte multiple");
  store_vec(reinterpret_cast<HB*>(dev) + PT::slot(), o);
  store_vec(reinterpret_cast<HB*>(host) + PT::slot(), o);
}

// grid = (njobs, nwin, nchunk), one wave: the jobs of k_treesum, slice z of the list on the wave's points
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_treesum_q(const typename PT::Point* __restrict__ segsum, const typename PT::Point* __restrict__ segw, uint32_t segs_per_win,
            HostPointOf<PT>* __restrict__ partial, int plain_job, typename PT::Point* __restrict__ stage,
            HostPointOf<PT>* __restrict__ partial_host) {
  const int job = blockIdx.x;
  const int w = blockIdx.y;
  const uint32_t nchunk = gridDim.z, z = blockIdx.z;
  const uint32_t quad = threadIdx.x / PT::LANES;
  const typename PT::Point* src = (job == 0 ? segw : segsum) + (size_t)w * segs_per_win;
  const bool whole = job == 0 || job == plain_job;
  const uint32_t len = whole ? segs_per_win : segs_per_win / 2;
  const uint32_t per = (len + nchunk - 1) / nchunk;
  const uint32_t lo = z * per, hi = lo + per < len ? lo + per : len;
  const uint32_t b = whole ? 0u : (uint32_t)(job - 1), lowmask = (1u << b) - 1u;
  PT acc = PT::infinity();
  for (uint32_t u = lo + quad; u < hi; u += PT::PER_WAVE) {
    // bit job: only the segments whose bit (job - 1) is set are enumerated
    const uint32_t t = whole ? u : (((u & ~lowmask) << 1) | (1u << b) | (u & lowmask));
    acc.add(PT::load(src + t));
  }
  acc = wave_quad_sum(acc);
  if (quad == 0) {
    const size_t idx = (size_t)w * gridDim.x + job;
    if (nchunk == 1) quad_store_host(acc, partial + idx, partial_host + idx);
    else acc.store(stage + idx * nchunk + z);
  }
}

// grid = (njobs, nwin): adds the nchunk slices of one (window, job)
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_treesum_final_q(const typename PT::Point* __restrict__ stage, uint32_t nchunk, HostPointOf<PT>* __restrict__ partial,
                  HostPointOf<PT>* __restrict__ partial_host) {
  const size_t idx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t quad = threadIdx.x / PT::LANES;
  PT acc = PT::infinity();
  for (uint32_t k = quad; k < nchunk; k += PT::PER_WAVE) acc.add(PT::load(stage + idx * nchunk + k));
  acc = wave_quad_sum(acc);
  if (quad == 0) quad_store_host(acc, partial + idx, partial_host + idx);
}

#ifdef ZKMI_EXPERIMENTS
// (A/B library only: measured in round 6, not adopted -- profiles/r06/experiments/quad_accumulation_ab.txt: at 4 - 8 lanes per
// bucket the launch fills the chip instead of being a latency chain, and an entry costs 16 products where the mixed addition
// needs 10: 2^14 unchanged, 2^16 2.85 -> 3.4 - 4.1 ms)
// Bucket ACCUMULATION with a point (quad / octet) per bucket, for ONE small proof or ONE small MSM by itself: the launch is a
// fraction of a millisecond of chip time but lasts as long as the fullest bucket's chain of dependent additions (17 entries
// per bucket at 2^14: 0.7 ms in the lane-pair G2 kernel, whose mixed addition is a chain of 10 Fq2 products); here an entry
// costs the 4 products of the quad form, the complete law handles P + P in line (no redo list), heavy buckets stay with the
// heavy kernels.  grid = (ceil(buckets / PER_WAVE), MSMs of the launch); the next entry is loaded before the current addition.
template <class PT>
struct AccumQArgs {
  const typename PT::APoint* bases[MSM_MULTI_MAX];
  typename PT::Point* buckets[MSM_MULTI_MAX];
  SortView sort[MSM_MULTI_MAX];
};
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_accum_q(const AccumQArgs<PT> args, uint32_t total_buckets) {
  const typename PT::APoint* __restrict__ const bases = args.bases[blockIdx.y];
  typename PT::Point* __restrict__ const buckets = args.buckets[blockIdx.y];
  const SortView& sv = args.sort[blockIdx.y];
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) / PT::LANES;
  if (t >= total_buckets) return;  // uniform per point
  const uint32_t b = sv.perm[t];
  const uint32_t cnt = sv.count[b];
  if (cnt > sv.heavy_thr) return;  // the heavy-bucket kernels own it
  const uint32_t beg = sv.begin[b], end = beg + cnt;
  PT acc = PT::infinity();
  if (beg < end) {
    uint32_t v = sv.sorted[beg];
    PT cur = PT::from_affine(bases + (v & 0x7fffffffu), (v >> 31) != 0);
    for (uint32_t j = beg + 1; j < end; j++) {
      v = sv.sorted[j];
      const PT nxt = PT::from_affine(bases + (v & 0x7fffffffu), (v >> 31) != 0);
      acc.add(cur);
      cur = nxt;
    }
    acc.add(cur);
  }
  acc.store(buckets + b);
}
#endif  // ZKMI_EXPERIMENTS

// k_accum_redo with a point (quad / octet) per stride of the listed bucket's entries (the entries enter as XYZZ points (x, +-y, 1, 1))
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_accum_redo_q(const typename PT::APoint* __restrict__ bases, const uint32_t* __restrict__ begin, const uint32_t* __restrict__ count,
               const uint32_t* __restrict__ sorted, typename PT::Point* __restrict__ buckets, uint32_t* __restrict__ redo,
               uint32_t* __restrict__ ticket, uint32_t into) {
  const uint32_t n = redo[0];
  const uint32_t quad = threadIdx.x / PT::LANES;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {  // wave-uniform
    const uint32_t b = redo[1 + k];
    const uint32_t beg = begin[b], end = beg + count[b];
    PT acc = PT::infinity();
    for (uint32_t j = beg + quad; j < end; j += PT::PER_WAVE) {
      const uint32_t v = sorted[j];
      acc.add(PT::from_affine(bases + (v & 0x7fffffffu), (v >> 31) != 0));
    }
    acc = wave_quad_sum(acc);
    if (quad == 0) {
      if (into) acc.add(PT::load(buckets + b));
      acc.store(buckets + b);
    }
  }
  // every workgroup has read the length by now; the last one to get here clears the list for the slot's next MSM
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    redo[0] = 0;
    *ticket = 0;
  }
}

// k_accum_heavy in quad / octet form, both modes (PARTIAL: the "points" of list entry h are the partial sums k_accum_heavy_nc
// left in pool_nc -- 64 per wave-item in G1, 32 in G2; POINT: the entries of sorted[], each as an XYZZ point): work items =
// (bucket, sub-range of ~4 points per quad) dealt round-robin to the one-wave workgroups of a 1-D grid; the last workgroup of a
// split bucket (ticket word) adds the slices.
template <class PT>
__global__ void __launch_bounds__(64, PT::LANES == 4 ? 3 : 2)
k_heavy_q(const typename PT::APoint* __restrict__ bases, const uint32_t* __restrict__ begin, const uint32_t* __restrict__ count,
          const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ sorted, typename PT::Point* __restrict__ buckets,
          typename PT::Point* __restrict__ heavy_partial, uint32_t* __restrict__ ticket, const uint32_t* __restrict__ hplan,
          const typename PT::Point* __restrict__ pool_nc) {
  constexpr uint32_t NCW = PT::LANES == 4 ? 64u : 32u;  // partial sums k_accum_heavy_nc[_g2] leaves per wave-item
  const bool pmode = hplan[3] == 0;
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t pl_nc = pmode ? hplan[1] : 0u;
  const uint32_t quad = threadIdx.x / PT::LANES;
  auto point_at = [&](uint32_t j) {
    const uint32_t sv = sorted[j];
    return PT::from_affine(bases + (sv & 0x7fffffffu), (sv >> 31) != 0);
  };
  // a partial sum, or -- where its lane (pair) met P + P (heavy_marker: zz = 0, word 0 of zzz[.c0] = 1) -- its points again
  auto partial_at = [&](uint32_t h, uint32_t j) {
    PT v = PT::load(pool_nc + j);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(v.v.l) / sizeof(v.v.l[0])); i++) o |= (uint32_t)v.v.l[i];
    const bool zz0 = PT::both(PT::template flag<QP_B2>(o == 0));
    const bool mk = PT::either(PT::template flag<0xFF>(v.v.l[0] == 1));  // lane 3: zzz (G2: the c0 lane carries the marker)
    if (zz0 && mk) {  // uniform per point
      const uint32_t it = j / NCW, ln = j % NCW;
      const uint32_t beg = ent[4 * h + 1], cnt = ent[4 * h + 2], first = ent[4 * h + 3];
      const uint32_t w0 = beg + (it - first) * NCW * pl_nc;
      const uint32_t w1 = (w0 + NCW * pl_nc < beg + cnt) ? w0 + NCW * pl_nc : beg + cnt;
      v = PT::infinity();
      for (uint32_t jj = w0 + ln; jj < w1; jj += NCW) v.add(point_at(jj));
    }
    return v;
  };
  const uint32_t n_heavy = pmode ? hplan[2] : heavy[0];
  const uint32_t n_wg = gridDim.x, wg = blockIdx.x;
  uint32_t item0 = 0;  // number of the bucket's first work item = its first pool slot (see k_accum_heavy)
  for (uint32_t h = 0; h < n_heavy; h++) {
    const uint32_t b = pmode ? ent[4 * h] : heavy[1 + h];
    const uint32_t cnt = pmode ? (ent[4 * (h + 1) + 3] - ent[4 * h + 3]) * NCW : count[b];
    const uint32_t beg0 = pmode ? ent[4 * h + 3] * NCW : begin[b];
    const uint32_t nsplit = heavy_nsplit(h, cnt, 4 * PT::PER_WAVE, item0);  // ~4 points per quad before the tree
    const uint32_t base = item0;
    const uint32_t r0 = (wg + n_wg - base % n_wg) % n_wg;
    item0 += nsplit;
    for (uint32_t r = r0; r < nsplit; r += n_wg) {  // wave-uniform
      uint32_t beg = beg0, end = beg0 + cnt;
      if (nsplit > 1) {
        const uint32_t len = (cnt + nsplit - 1) / nsplit;
        const uint32_t sb = beg + r * len;
        end = (sb + len < end) ? sb + len : end;
        beg = sb < end ? sb : end;
      }
      PT acc = PT::infinity();
      if (pmode) {
        for (uint32_t j = beg + quad; j < end; j += PT::PER_WAVE) acc.add(partial_at(h, j));
      } else {
        for (uint32_t j = beg + quad; j < end; j += PT::PER_WAVE) acc.add(point_at(j));
      }
      acc = wave_quad_sum(acc);
      if (nsplit == 1) {
        if (quad == 0) acc.store(buckets + b);
      } else {
        if (quad == 0) acc.store(heavy_partial + (size_t)base + r);
        __threadfence();  // the lanes' parts of the partial are visible device-wide before the ticket is taken
        uint32_t last = 0;
        if (threadIdx.x == 0) last = atomicAdd(ticket + h, 1u) == nsplit - 1 ? 1u : 0u;
        last = (uint32_t)__shfl((int)last, 0);
        if (last) {  // wave-uniform
          __threadfence();
          PT v = PT::infinity();
          for (uint32_t k = quad; k < nsplit; k += PT::PER_WAVE) v.add(PT::load(heavy_partial + (size_t)base + k));
          v = wave_quad_sum(v);
          if (quad == 0) v.store(buckets + b);
          if (threadIdx.x == 0) ticket[h] = 0;
        }
      }
    }
  }
}

}  // namespace zkmi

// zkmi — the G1 reduction-side kernels of the MSM with every complete addition split over a lane quad (quad.hpp):
// segment sums, tree sums, the redo pass and the summing of heavy-bucket partials.  Included by msm_impl.hpp.
//
// All of them are ONE-WAVE workgroups (16 quads) of at most 168 registers without scratch or LDS -- the shape of the
// accumulation kernels -- so they are placed in the slot a retiring accumulation wave frees instead of waiting for the
// accumulation grids to drain (the one-lane forms: 230-320 registers, 128/256-thread workgroups), and a
// dependent addition is a chain of 4 field products instead of 14.
#pragma once
#include "quad.hpp"

namespace zkmi {

constexpr uint32_t MSM_NQ = 16;  // quads per wave

// quad t handles buckets [t*seg, (t+1)*seg); buckets2 / buckets3: see k_segreduce
template <class F>
__global__ void __launch_bounds__(64, 3)
k_segreduce_q(const XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ segsum, XYZZ<F>* __restrict__ segw,
              uint32_t total_segs, int seg, const XYZZ<F>* __restrict__ buckets2, const XYZZ<F>* __restrict__ buckets3) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (t >= total_segs) return;  // quad-uniform
  XYZZQ<F> run = XYZZQ<F>::infinity();
  XYZZQ<F> acc = XYZZQ<F>::infinity();
  for (int i = seg - 1; i >= 0; i--) {
    run.add(XYZZQ<F>::load(buckets + (size_t)t * seg + i));
    if (buckets2) run.add(XYZZQ<F>::load(buckets2 + (size_t)t * seg + i));
    if (buckets3) run.add(XYZZQ<F>::load(buckets3 + (size_t)t * seg + i));
    acc.add(run);
  }
  run.store(segsum + t);
  acc.store(segw + t);
}

// lane q of quad 0 converts coordinate q to the host representation and writes it to the device copy and the pinned host slot
template <class F>
__device__ __forceinline__ void quad_store_host(const XYZZQ<F>& acc, XYZZ<typename HostFieldOf<F>::type>* dev,
                                                XYZZ<typename HostFieldOf<F>::type>* host) {
  using HF = typename HostFieldOf<F>::type;
  static_assert(sizeof(HF) % 16 == 0, "16-byte multiple");
  const HF o = fq_from_fq28(acc.v);
  store_vec(reinterpret_cast<HF*>(dev) + XYZZQ<F>::q(), o);
  store_vec(reinterpret_cast<HF*>(host) + XYZZQ<F>::q(), o);
}

// grid = (njobs, nwin, nchunk), one wave: the jobs of k_treesum, slice z of the list on 16 quads
template <class F>
__global__ void __launch_bounds__(64, 3)
k_treesum_q(const XYZZ<F>* __restrict__ segsum, const XYZZ<F>* __restrict__ segw, uint32_t segs_per_win,
            XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial, int plain_job, XYZZ<F>* __restrict__ stage,
            XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial_host) {
  const int job = blockIdx.x;
  const int w = blockIdx.y;
  const uint32_t nchunk = gridDim.z, z = blockIdx.z;
  const uint32_t quad = threadIdx.x >> 2;
  const XYZZ<F>* src = (job == 0 ? segw : segsum) + (size_t)w * segs_per_win;
  const bool whole = job == 0 || job == plain_job;
  const uint32_t len = whole ? segs_per_win : segs_per_win / 2;
  const uint32_t per = (len + nchunk - 1) / nchunk;
  const uint32_t lo = z * per, hi = lo + per < len ? lo + per : len;
  const uint32_t b = whole ? 0u : (uint32_t)(job - 1), lowmask = (1u << b) - 1u;
  XYZZQ<F> acc = XYZZQ<F>::infinity();
  for (uint32_t u = lo + quad; u < hi; u += MSM_NQ) {
    // bit job: only the segments whose bit (job - 1) is set are enumerated
    const uint32_t t = whole ? u : (((u & ~lowmask) << 1) | (1u << b) | (u & lowmask));
    acc.add(XYZZQ<F>::load(src + t));
  }
  acc = wave_quad_sum(acc);
  if (quad == 0) {
    const size_t idx = (size_t)w * gridDim.x + job;
    if (nchunk == 1) quad_store_host(acc, partial + idx, partial_host + idx);
    else acc.store(stage + idx * nchunk + z);
  }
}

// grid = (njobs, nwin): adds the nchunk slices of one (window, job)
template <class F>
__global__ void __launch_bounds__(64, 3)
k_treesum_final_q(const XYZZ<F>* __restrict__ stage, uint32_t nchunk, XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial,
                  XYZZ<typename HostFieldOf<F>::type>* __restrict__ partial_host) {
  const size_t idx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t quad = threadIdx.x >> 2;
  XYZZQ<F> acc = XYZZQ<F>::infinity();
  for (uint32_t k = quad; k < nchunk; k += MSM_NQ) acc.add(XYZZQ<F>::load(stage + idx * nchunk + k));
  acc = wave_quad_sum(acc);
  if (quad == 0) quad_store_host(acc, partial + idx, partial_host + idx);
}

// k_accum_redo with a quad per stride of the listed bucket's entries (the entries enter as XYZZ points (x, +-y, 1, 1))
template <class F>
__global__ void __launch_bounds__(64, 3)
k_accum_redo_q(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin, const uint32_t* __restrict__ count,
               const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets, uint32_t* __restrict__ redo,
               uint32_t* __restrict__ ticket, uint32_t into) {
  const uint32_t n = redo[0];
  const uint32_t quad = threadIdx.x >> 2;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {  // wave-uniform
    const uint32_t b = redo[1 + k];
    const uint32_t beg = begin[b], end = beg + count[b];
    XYZZQ<F> acc = XYZZQ<F>::infinity();
    for (uint32_t j = beg + quad; j < end; j += MSM_NQ) {
      const uint32_t v = sorted[j];
      acc.add(XYZZQ<F>::from_affine(bases + (v & 0x7fffffffu), (v >> 31) != 0));
    }
    acc = wave_quad_sum(acc);
    if (quad == 0) {
      if (into) acc.add(XYZZQ<F>::load(buckets + b));
      acc.store(buckets + b);
    }
  }
  // every workgroup has read the length by now; the last one to get here clears the list for the slot's next MSM
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    redo[0] = 0;
    *ticket = 0;
  }
}

// k_accum_heavy in quad form, both modes (PARTIAL: the "points" of list entry h are the partial sums k_accum_heavy_nc left in
// pool_nc; POINT: the entries of sorted[], each as an XYZZ point): work items = (bucket, sub-range of 64 points) dealt
// round-robin to the one-wave workgroups of a 1-D grid; the last workgroup of a split bucket (ticket word) adds the slices.
template <class F>
__global__ void __launch_bounds__(64, 3)
k_heavy_q(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin, const uint32_t* __restrict__ count,
          const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets,
          XYZZ<F>* __restrict__ heavy_partial, uint32_t* __restrict__ ticket, const uint32_t* __restrict__ hplan,
          const XYZZ<F>* __restrict__ pool_nc) {
  const bool pmode = hplan[3] == 0;
  const uint32_t* __restrict__ const ent = hplan + 4;
  const uint32_t pl_nc = pmode ? hplan[1] : 0u;
  const uint32_t quad = threadIdx.x >> 2;
  auto point_at = [&](uint32_t j) {
    const uint32_t sv = sorted[j];
    return XYZZQ<F>::from_affine(bases + (sv & 0x7fffffffu), (sv >> 31) != 0);
  };
  // a partial sum, or -- where its lane met P + P (heavy_marker: zz = 0, zzz word 0 = 1) -- the lane's points again
  auto partial_at = [&](uint32_t h, uint32_t j) {
    XYZZQ<F> v = XYZZQ<F>::load(pool_nc + j);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F::NL; i++) o |= (uint32_t)v.v.l[i];
    const bool zz0 = quad_flag<QP_B2>(o == 0);
    const bool mk = __builtin_amdgcn_mov_dpp(v.v.l[0] == 1 ? 1 : 0, 0xFF, 0xF, 0xF, true) != 0;  // lane 3: zzz
    if (zz0 && mk) {  // quad-uniform
      const uint32_t it = j >> 6, ln = j & 63u;
      const uint32_t beg = ent[4 * h + 1], cnt = ent[4 * h + 2], first = ent[4 * h + 3];
      const uint32_t w0 = beg + (it - first) * 64u * pl_nc;
      const uint32_t w1 = (w0 + 64u * pl_nc < beg + cnt) ? w0 + 64u * pl_nc : beg + cnt;
      v = XYZZQ<F>::infinity();
      for (uint32_t jj = w0 + ln; jj < w1; jj += 64) v.add(point_at(jj));
    }
    return v;
  };
  const uint32_t n_heavy = pmode ? hplan[2] : heavy[0];
  const uint32_t n_wg = gridDim.x, wg = blockIdx.x;
  uint32_t item0 = 0;  // number of the bucket's first work item = its first pool slot (see k_accum_heavy)
  for (uint32_t h = 0; h < n_heavy; h++) {
    const uint32_t b = pmode ? ent[4 * h] : heavy[1 + h];
    const uint32_t cnt = pmode ? (ent[4 * (h + 1) + 3] - ent[4 * h + 3]) * 64u : count[b];
    const uint32_t beg0 = pmode ? ent[4 * h + 3] * 64u : begin[b];
    const uint32_t nsplit = heavy_nsplit(h, cnt, 4 * MSM_NQ, item0);  // ~4 points per quad before the tree
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
      XYZZQ<F> acc = XYZZQ<F>::infinity();
      if (pmode) {
        for (uint32_t j = beg + quad; j < end; j += MSM_NQ) acc.add(partial_at(h, j));
      } else {
        for (uint32_t j = beg + quad; j < end; j += MSM_NQ) acc.add(point_at(j));
      }
      acc = wave_quad_sum(acc);
      if (nsplit == 1) {
        if (quad == 0) acc.store(buckets + b);
      } else {
        if (quad == 0) acc.store(heavy_partial + (size_t)base + r);
        __threadfence();  // the four lanes' parts of the partial are visible device-wide before the ticket is taken
        uint32_t last = 0;
        if (threadIdx.x == 0) last = atomicAdd(ticket + h, 1u) == nsplit - 1 ? 1u : 0u;
        last = (uint32_t)__shfl((int)last, 0);
        if (last) {  // wave-uniform
          __threadfence();
          XYZZQ<F> v = XYZZQ<F>::infinity();
          for (uint32_t k = quad; k < nsplit; k += MSM_NQ) v.add(XYZZQ<F>::load(heavy_partial + (size_t)base + k));
          v = wave_quad_sum(v);
          if (quad == 0) v.store(buckets + b);
          if (threadIdx.x == 0) ticket[h] = 0;
        }
      }
    }
  }
}

}  // namespace zkmi

// zkmi — A/B LIBRARY ONLY (-DZKMI_EXPERIMENTS): the retired generations of the MSM accumulation kernels that the switches of
// tune.hpp still select in libzkmi_exp.so -- the first thread-per-bucket kernels with an out-of-line doubling path (ZKMI_ACCUM
// = 0 / 1), the first lane-pair G2 kernel (ZKMI_ACCUM_G2 = 0), and the occupancy-capped two-wave build of the call-free G1
// kernel (the pipelined big MSM, msm_pipe.hpp: measured neutral).  Nothing here is compiled into the product library; the
// variants test (tests/test_gpu_sizes.py::test_ab_switches_do_not_change_any_result) holds every one of them to its bytes.
// Included by msm_impl.hpp behind the definitions it uses (load_vec / store_vec, ld_comp / st_comp, AccumArgs, madd_generic).
#pragma once

namespace zkmi {

// ---- retired accumulation kernels (A/B library only: ZKMI_ACCUM=0|1, ZKMI_ACCUM_G2=0|1) ----
// G1 (14-limb coordinates): 248 VGPRs -> 2 waves per SIMD.  G2 needs ~330 registers
// (accumulator 112 + point 56 + columns 56 + temporaries) and runs at 1 wave per SIMD with
// cheap AGPR spills; forcing 2 waves sends 350+ values to scratch and is 2x slower.
template <class F>
struct AccumWaves {
  static constexpr int value = (sizeof(F) <= 64) ? 2 : 1;  // measured: forcing 3 for G1 spills around the rare-path calls and is 25 % slower
};
template <class F>
__global__ void __launch_bounds__(256, AccumWaves<F>::value)
k_accum(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin,
        const uint32_t* __restrict__ count, const uint32_t* __restrict__ perm,
        const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets, uint32_t total_buckets,
        uint32_t heavy_thr) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_buckets) return;
  const uint32_t b = perm[t];  // lanes of a wave own buckets of near-equal load
  const uint32_t cnt = count[b];
  if (cnt > heavy_thr) return;  // k_accum_heavy owns it
  const uint32_t beg = begin[b], end = beg + cnt;
  XYZZ<F> acc = XYZZ<F>::infinity();
  // (prefetching the next point into registers costs 28 VGPRs and gains nothing; the G1 path of
  // the prover prefetches through LDS instead: k_accum_g1_glds below)
  for (uint32_t j = beg; j < end; j++) {
    const uint32_t v = sorted[j];
    Affine<F> p = load_vec(bases + (v & 0x7fffffffu));
    if (v >> 31) p.y = p.y.neg();
    acc.madd(p);
  }
  store_vec(buckets + b, acc);
}

// G1 accumulation with the gather of the NEXT point in flight during the current mixed addition,
// at no register cost: the 112-byte (BLS12-381) or 80-byte (BN254) table entry is fetched by seven direct-to-LDS loads
// (global_load_lds_dwordx4: per-lane source address, destination = wave-uniform LDS base + 16 B x lane),
// into one of two LDS buffers.  Order inside an iteration: wait -> read point j from LDS -> issue the
// loads of point j+1 and of index j+2 -> mixed addition (no memory operation inside it).
// LDS: 2 buffers x 256 threads x 112 B = 56 KB per block, two blocks per CU (VGPR-limited anyway).
// BW = waves per workgroup.  One-wave workgroups (BW = 1) free their slot the moment the wave retires; a
// 4-wave workgroup can only start once all four SIMDs of a CU have a free slot at the same time.
// The loop strides the load-ordered bucket list by the grid size: with a grid of one wave per bucket group
// it runs once (dynamic dispatch, heaviest groups first); with a smaller grid every wave takes a heavy, a
// medium and a light group in turn (ZKMI_ACCUM_ROUNDS) and nothing depends on the dispatcher's refill rate.
template <class F, int BW>
__global__ void __launch_bounds__(64 * BW, AccumWaves<F>::value)
k_accum_g1_glds(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ begin,
                const uint32_t* __restrict__ count, const uint32_t* __restrict__ perm,
                const uint32_t* __restrict__ sorted, XYZZ<F>* __restrict__ buckets, uint32_t total_buckets,
                uint32_t heavy_thr) {
  constexpr int CHUNKS = sizeof(Affine<F>) / 16;  // 7 (BLS12-381 Fq), 5 (BN254 Fq)
  __shared__ uint4 tile[2][BW][CHUNKS][64];           // [buffer][wave][chunk][lane]
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total_buckets; t += stride) {
    const uint32_t b = perm[t];  // lanes of a wave own buckets of near-equal load
    const uint32_t cnt = count[b];
    if (cnt > heavy_thr) continue;  // k_accum_heavy owns it
    const uint32_t beg = begin[b], end = beg + cnt;
    XYZZ<F> acc = XYZZ<F>::infinity();
    auto fetch = [&](uint32_t v, int buf) {
      const char* src = reinterpret_cast<const char*>(bases + (v & 0x7fffffffu));
#pragma unroll
      for (int q = 0; q < CHUNKS; q++)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 16 * q), (lds_ptr_t)&tile[buf][wave][q][0], 16, 0, 0);
    };
    uint32_t v_cur = 0, v_next = 0;
    if (cnt) {
      v_cur = sorted[beg];
      fetch(v_cur, 0);
      if (cnt > 1) v_next = sorted[beg + 1];
    }
    int buf = 0;
    for (uint32_t j = beg; j < end; j++) {
      // the compiler waits for the outstanding LDS-DMA (vmcnt) before these LDS reads
      Affine<F> p;
      uint4* d = reinterpret_cast<uint4*>(&p);
#pragma unroll
      for (int q = 0; q < CHUNKS; q++) d[q] = tile[buf][wave][q][lane];
      const uint32_t v = v_cur;
      if (j + 1 < end) {
        fetch(v_next, buf ^ 1);
        v_cur = v_next;
        if (j + 2 < end) v_next = sorted[j + 2];
      }
      buf ^= 1;
      if (v >> 31) p.y = p.y.neg();
      acc.madd(p);
    }
    store_vec(buckets + b, acc);
  }
}

template <int BW>
__global__ void __launch_bounds__(64 * BW, 2)
k_accum_g2_split(const Affine<Fq2_28>* __restrict__ bases, const uint32_t* __restrict__ begin,
                 const uint32_t* __restrict__ count, const uint32_t* __restrict__ perm,
                 const uint32_t* __restrict__ sorted, XYZZ<Fq2_28>* __restrict__ buckets, uint32_t total_buckets,
                 uint32_t heavy_thr) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x; (gt >> 1) < total_buckets; gt += stride) {
    const uint32_t t = gt >> 1, comp = gt & 1u;
    const uint32_t b = perm[t];
    const uint32_t cnt = count[b];
    if (cnt > heavy_thr) continue;  // pair-uniform
    const uint32_t beg = begin[b], end = beg + cnt;
    XYZZ<Fq2P> acc = XYZZ<Fq2P>::infinity();
    for (uint32_t j = beg; j < end; j++) {
      const uint32_t v = sorted[j];
      const Fq28* src = reinterpret_cast<const Fq28*>(bases + (v & 0x7fffffffu));  // x.c0 x.c1 y.c0 y.c1
      Affine<Fq2P> p;
      p.x.v = ld_comp(src + comp);
      p.y.v = ld_comp(src + 2 + comp);
      if (v >> 31) p.y = p.y.neg();
      acc.madd(p);
    }
    Fq28* dst = reinterpret_cast<Fq28*>(buckets + b);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
    st_comp(dst + comp, acc.x.v);
    st_comp(dst + 2 + comp, acc.y.v);
    st_comp(dst + 4 + comp, acc.zz.v);
    st_comp(dst + 6 + comp, acc.zzz.v);
  }
}

// The same loop once more, as a function, for the occupancy-capped kernel below.  (NOT shared with k_accum_g1_nc: routed
// through a function the three-wave kernel spills 92 bytes per lane -- its zero spills at exactly 168 registers are a draw
// of the register allocator that any change of the surrounding code loses: DESIGN.md section 10.)
template <class F, int BW, bool MULTI, bool INTO>
__device__ __forceinline__ void accum_g1_nc_body(const AccumArgs<F, MULTI>& args, uint32_t total_buckets) {
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
// The same loop CAPPED at two waves per SIMD with at most 176 registers (352 of a SIMD's 512: the kernel descriptor is
// padded to the occupancy limit): it leaves 160 registers per SIMD, ~100 KB of LDS per CU and six wave slots per SIMD to
// OTHER kernels.  The three-wave kernel above fills 504 registers and nothing runs beside it; the
// issue rate of the additions is the same at two and at three waves (round 2).  Used by the pipelined form of one big
// windowed MSM (BASELINE config 3): the digit sort of the second window group runs beside the first group's accumulation.
template <class F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(176)))
k_accum_g1_nc_w2(const AccumArgs<F, false> args, uint32_t total_buckets) {
  accum_g1_nc_body<F, 1, false, false>(args, total_buckets);
}

}  // namespace zkmi

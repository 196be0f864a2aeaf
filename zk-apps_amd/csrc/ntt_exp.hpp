// zkmi — A/B LIBRARY ONLY (-DZKMI_EXPERIMENTS): the register-blocked and one-wave NTT pass kernels (ZKMI_NTT_RB = 1, 2, 4, 5),
// measured in rounds 3 and 5 and not adopted (DESIGN.md sections 5.2, 10); byte-identical to the product's k_ntt_pass
// (tests/test_gpu_sizes.py::test_ab_switches_do_not_change_any_result).  Textually included by ntt.hip inside its namespaces,
// behind ld28 / st28 / st_words8 and k_ntt_pass.
// ---------------------------------------------------------------------------------------------
// (A/B library only: measured in round 3, not adopted -- DESIGN.md 4.2)
// Register-blocked pass (round 3).  Same tiles, same twists and the same stage order as k_ntt_pass, but a thread
// owns E = 2^LOGE tile elements per ROUND and runs up to LOGE butterfly stages on them in registers (radix-8
// sub-butterflies for LOGE = 3: 12 products between two barriers instead of one).  A 2048-element tile is a
// 256-thread workgroup: 10 stages = 4 barriers instead of 10, 3.3x fewer LDS round trips per butterfly, and -- what
// matters inside the prover -- one wave per SIMD instead of four, so the workgroup fits beside the bucket
// accumulation's waves (168 VGPRs x 3 per SIMD) as soon as ONE of them retires; the 1024-thread form needed a
// whole CU to drain (0.49 ms per pass in the round-2 pipeline trace against 0.14 ms alone).
// LDS layout: limb-major 32-bit words, element index padded by one word per 32 (index L -> L + L / 32), so that the
// strided element sets of a round (stride 2^(u0 + Q) elements between a thread's own elements, stride 1 or 2^K
// between lanes) fall into distinct banks; the twiddles of the sub-transform use the same layout.
// compile-time loops: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- the register
// arrays of the blocked pass are only ever indexed with constants, so they stay in VGPRs
template <int... Is, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

template <class F>
struct RbLds {
  uint32_t* w;
  uint32_t pitch;
  __device__ __forceinline__ static uint32_t pad(uint32_t L) { return L + (L >> 5); }
  __device__ __forceinline__ F ld(uint32_t L) const {
    F r;
    const uint32_t o = pad(L);
#pragma unroll
    for (int i = 0; i < F::NL; i++) r.l[i] = (int32_t)w[i * pitch + o];
    return r;
  }
  __device__ __forceinline__ void st(uint32_t L, const F& v) const {
    const uint32_t o = pad(L);
#pragma unroll
    for (int i = 0; i < F::NL; i++) w[i * pitch + o] = (uint32_t)v.l[i];
  }
};

// LOGR > 0: a thread owns 2^LOGR units of E elements per round and works through them one after the other -- the
// one-wave form (LOGE = 2, LOGR = 2 on a 1024-element tile: 64 threads, <= 160 VGPRs, 63 KB of LDS) is the only NTT
// workgroup that finds a place beside the bucket accumulations.
// ONEW (LOGE = 2, LOGR = 1 on a 512-element tile): the whole workgroup is one wave of <= 170 VGPRs with 24 KB of LDS -- up
// to three of them fit a CU beside twelve accumulation waves (mode 5 below: three passes of <= 7 stages at N = 2^20).
// (Eight elements per thread would halve the LDS round trips again but needs 247 VGPRs: 75 of them spill at 168.)
template <class F, bool DIF, bool LOCAL_TW, int LOGE, int LOGR = 0, bool ONEW = false>
__global__ void __launch_bounds__((LOGR || ONEW) ? 64 : (2048 >> LOGE), ONEW ? 3 : 1)
k_ntt_pass_rb(F* __restrict__ data, const F* __restrict__ tw, int log_n, int t0, int S, int Q,
              const F* __restrict__ post, uint32_t* __restrict__ canon_out) {
  constexpr int E = 1 << LOGE;
  constexpr int REP = 1 << LOGR;
  data += (size_t)blockIdx.y << log_n;
  if (canon_out) canon_out += ((size_t)blockIdx.y << log_n) * 8;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const uint32_t tile_n = 1u << (S + Q);
  const uint32_t nthr = tile_n >> (LOGE + LOGR);  // == blockDim.x
  const uint32_t nunit = tile_n >> LOGE;        // units of E elements per round
  RbLds<F> tile{reinterpret_cast<uint32_t*>(lds_raw), tile_n + (tile_n >> 5)};
  RbLds<F> ctw{tile.w + (size_t)F::NL * tile.pitch, (1u << (S - 1)) + ((1u << (S - 1)) >> 5) + 1u};
  const uint32_t blk = blockIdx.x, tid = threadIdx.x;
  const uint32_t mid_bits = (t0 > 0) ? (uint32_t)(t0 - Q) : 0u;
  const uint32_t mid = blk & ((1u << mid_bits) - 1u);
  const uint32_t hi = blk >> mid_bits;
  const uint32_t qeff = (t0 > 0) ? (uint32_t)Q : 0u;
  auto gindex = [&](uint32_t L) -> uint32_t {
    if (t0 == 0) return blk * tile_n + L;
    const uint32_t e = L >> Q, c = L & ((1u << Q) - 1u);
    return (hi << (t0 + S)) | (e << t0) | (mid << Q) | c;
  };
  auto twist = [&](uint32_t L) -> F {
    const uint32_t e = L >> Q, c = L & ((1u << Q) - 1u);
    const uint32_t col = (mid << Q) | c;
    // stages [0, t0 + S) are independent 2^(t0+S)-point transforms: the cross terms are powers of the root of THAT order,
    // w_N^(2^(log_n - t0 - S)) -- the shift is zero for the last pass of a plan, non-zero for a middle pass
    const uint32_t ex = (col * (__brev(e) >> (32 - S))) << (log_n - t0 - S);
    const uint32_t halfn = 1u << (log_n - 1);
    F f = ld28(tw + (ex & (halfn - 1u)));
    return (ex & halfn) ? f.neg() : f;
  };
  if (LOCAL_TW)
    for (uint32_t k = tid; k < (1u << (S - 1)); k += nthr) ctw.st(k, ld28(tw + ((size_t)k << (log_n - S))));
#pragma unroll
  for (int m = 0; m < E * REP; m++) {
    const uint32_t L = tid + (uint32_t)m * nthr;
    F v = ld28(data + gindex(L));
    if (LOCAL_TW && !DIF && t0 > 0) v = v * twist(L);
    tile.st(L, v);
  }
  __syncthreads();

  // one round = K consecutive stages [u0, u0 + K) on E elements per thread (2^(LOGE - K) independent groups of 2^K)
  auto round = [&](auto kc, int u0) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    const uint32_t sh = (uint32_t)u0 + qeff;
    for (uint32_t unit = tid; unit < nunit; unit += nthr) {
    F x[E];
    uint32_t Lm[E];
    static_for<E>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr uint32_t j = (uint32_t)m & ((1u << K) - 1u), g = (uint32_t)m >> K;
      const uint32_t G = (unit << (LOGE - K)) | g;
      Lm[m] = ((G >> sh) << (sh + K)) | (G & ((1u << sh) - 1u)) | (j << sh);
      x[m] = tile.ld(Lm[m]);
    });
    static_for<K>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = DIF ? (K - 1 - decltype(ic)::value) : decltype(ic)::value;
      const int u = u0 + i;
      const uint32_t dist_log = (uint32_t)u + qeff;
      static_for<E>([&](auto m0c) __attribute__((always_inline)) {
        constexpr int m0 = decltype(m0c)::value;
        if constexpr ((m0 & (1 << i)) == 0) {
          constexpr int m1 = m0 | (1 << i);
          F w;
          if (LOCAL_TW) {
            const uint32_t lo = Lm[m0] & ((1u << dist_log) - 1u);
            w = ctw.ld((lo >> qeff) << (S - 1 - u));
          } else {
            const int t = t0 + u;
            const uint32_t jj = gindex(Lm[m0]) & ((1u << t) - 1u);
            w = ld28(tw + ((size_t)jj << (log_n - 1 - t)));
          }
          if (DIF) {
            const F a = x[m0], b = x[m1];
            x[m0] = a + b;
            x[m1] = a.sub_lazy(b) * w;
          } else {
            const F y = x[m1] * w;
            const F a = x[m0];
            x[m0] = a + y;
            x[m1] = a - y;
          }
        }
      });
    });
    static_for<E>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      tile.st(Lm[m], x[m]);
    });
    }  // units of this thread (disjoint element sets: no barrier between them)
    __syncthreads();
  };
  {
    int done = 0;
    while (done < S) {
      int k = S - done;
      if (k > LOGE) k = LOGE;
      const int u0 = DIF ? (S - done - k) : done;
      if (LOGE >= 3 && k == 3) round(std::integral_constant<int, (LOGE >= 3 ? 3 : 1)>{}, u0);
      else if (LOGE >= 2 && k == 2) round(std::integral_constant<int, (LOGE >= 2 ? 2 : 1)>{}, u0);
      else round(std::integral_constant<int, 1>{}, u0);
      done += k;
    }
  }

#pragma unroll
  for (int m = 0; m < E * REP; m++) {
    const uint32_t L = tid + (uint32_t)m * nthr;
    const uint32_t g = gindex(L);
    F v = tile.ld(L);
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

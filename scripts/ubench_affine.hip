// What would a bucket addition cost in AFFINE coordinates with the inversion shared by a whole wave?  
//
// The G1 bucket accumulation spends 4 550-4 670 wave-instructions per XYZZ mixed addition (8 products + 2 squares, one of
// the products double: section 3), and the accumulations are 83 % of a proof's instructions.  The affine chord rule needs
// one inversion per addition; Montgomery's trick turns n inversions into 3 (n - 1) products and one inversion, and on a
// SIMD machine the one inversion costs the same whether one lane or sixty-four compute it -- so the lanes of a wave share
// it: every lane multiplies up the denominators of its own NPL independent additions (prefix products kept in HBM: the
// accumulations leave it idle), the 64 lane products are combined by two log-step scans over the wave (12 products +
// shuffles), every lane inverts the SAME total (381 squarings + ~190 products, Fermat), and each lane unwinds its batch:
//     per addition   1 (prefix)  +  2 (unwind)  +  1 (lambda)  +  1 square  +  1 (y3)      =  5 products + 1 square
//     per lane batch 12 + 2 products for the scans, per wave batch one inversion
// This program measures that on independent PAIRS (out[q] = in[2 q] + in[2 q + 1], the first level of a pairwise tree over a
// sorted bucket list) against the XYZZ mixed addition on the same pairs with the same memory pattern, checks every affine
// result against the XYZZ one (x3 zz = X, y3 zzz = Y), and reports time per addition; scripts/ubench_affine.sh adds the
// instruction counts (rocprofv3 --pmc SQ_INSTS_VALU).  Operands are random field elements: the chord rule does not involve
// the curve equation.  Equal abscissas (doubling / cancellation) are flagged and skipped, as a product kernel would hand
// them to the complete addition.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-apps_amd/csrc scripts/ubench_affine.hip -o scripts/_bin/ubench_affine
// Usage: ubench_affine [log2 pairs = 22] [repeats = 5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "curve.hpp"
#include "field28.hpp"
#include "msm_impl.hpp"

using namespace zkmi;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ Fq28 seed_fq(uint64_t s) {
  Fq28 r;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    r.l[i] = (int32_t)(s >> 33) & Fq28::MASK;
  }
  r.l[Fq28::NL - 1] &= 0x3fff;  // below the modulus' top limb
  return r;
}

__global__ void k_fill(Affine<Fq28>* pts, uint64_t n, uint64_t seed, uint32_t n_equal) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<Fq28> p = {seed_fq(seed + 2 * i), seed_fq(seed * 7 + 2 * i + 1)};
  // a few pairs with equal abscissas (every 2^k-th pair): the flagged path
  if (n_equal && (i & 1) && ((i >> 1) % n_equal) == 5) p.x = seed_fq(seed + 2 * (i - 1));
  pts[i] = p;
}

__device__ __forceinline__ Fq28 shfl_from(const Fq28& v, int src) {
  Fq28 r;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) r.l[i] = __shfl(v.l[i], src, 64);
  return r;
}
__device__ __forceinline__ Fq28 select(bool c, const Fq28& a, const Fq28& b) {
  Fq28 r;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// out[q] = in[2 q] + in[2 q + 1] (affine), NPL pairs per lane, one inversion per wave
template <int NPL, int W>
__global__ void __launch_bounds__(64, W)
k_pair_affine(const Affine<Fq28>* __restrict__ in, Affine<Fq28>* __restrict__ out, Fq28* __restrict__ scratch, uint32_t* __restrict__ flagged) {
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 64 * NPL;
  Fq28 acc = Fq28::one();
  uint64_t equal_mask[(NPL + 63) / 64] = {};  // equal abscissas: found once, on the way up
  for (int i = 0; i < NPL; i++) {
    const size_t q = base + (size_t)i * 64 + lane;
    Fq28 d = in[2 * q + 1].x - in[2 * q].x;
    if (d.is_zero()) {
      d = Fq28::one();
      equal_mask[i >> 6] |= 1ull << (i & 63);
    }
    acc = acc * d;
    scratch[q] = acc;
  }
  // the lane products combined: exclusive prefix and exclusive suffix over the wave
  Fq28 pre = acc, suf = acc;
#pragma unroll 1
  for (int s = 1; s < 64; s <<= 1) {
    const Fq28 t = shfl_from(pre, lane >= s ? lane - s : lane);
    pre = select(lane >= s, pre * t, pre);
  }
#pragma unroll 1
  for (int s = 1; s < 64; s <<= 1) {
    const Fq28 t = shfl_from(suf, lane + s < 64 ? lane + s : lane);
    suf = select(lane + s < 64, suf * t, suf);
  }
  const Fq28 total = shfl_from(pre, 63);
  Fq28 left = shfl_from(pre, lane > 0 ? lane - 1 : 0), right = shfl_from(suf, lane < 63 ? lane + 1 : 63);
  left = select(lane > 0, left, Fq28::one());
  right = select(lane < 63, right, Fq28::one());
  Fq28 inv_acc = total.inv() * left * right;  // 1 / (this lane's product)
  for (int i = NPL - 1; i >= 0; i--) {
    const size_t q = base + (size_t)i * 64 + lane;
    const Affine<Fq28> p1 = in[2 * q], p2 = in[2 * q + 1];
    const bool equal = (equal_mask[i >> 6] >> (i & 63)) & 1;
    const Fq28 d = equal ? Fq28::one() : p2.x.sub_lazy(p1.x);
    Fq28 inv_d = inv_acc;
    if (i > 0) {
      inv_d = inv_acc * scratch[q - 64];
      inv_acc = inv_acc * d;
    }
    if (equal) {
      atomicAdd(flagged, 1u);
      continue;
    }
    const Fq28 lam = p2.y.sub_lazy(p1.y) * inv_d;
    Fq28 x3 = lam.sqr();
#pragma unroll
    for (int k = 0; k < Fq28::NL; k++) x3.l[k] -= p1.x.l[k] + p2.x.l[k];
    x3.carry();
    Fq28 y3 = lam * p1.x.sub_lazy(x3);
#pragma unroll
    for (int k = 0; k < Fq28::NL; k++) y3.l[k] -= p1.y.l[k];
    y3.carry();
    out[q] = {x3, y3};
  }
}

// the same pairs through the XYZZ mixed addition (what k_accum_g1_nc does per sorted entry)
template <int W>
__global__ void __launch_bounds__(64, W)
k_pair_xyzz(const Affine<Fq28>* __restrict__ in, XYZZ<Fq28>* __restrict__ out, uint32_t* __restrict__ flagged) {
  const size_t q = (size_t)blockIdx.x * 64 + threadIdx.x;
  const Affine<Fq28> p1 = in[2 * q], p2 = in[2 * q + 1];
  XYZZ<Fq28> acc = {p1.x, p1.y, Fq28::one(), Fq28::one()};
  if (!madd_generic(acc, p2, 0u)) {
    atomicAdd(flagged, 1u);
    return;
  }
  out[q] = acc;
}

__global__ void k_compare(const Affine<Fq28>* __restrict__ in, const Affine<Fq28>* __restrict__ aff, const XYZZ<Fq28>* __restrict__ xy, uint64_t n,
                          uint32_t* __restrict__ bad) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  if ((in[2 * q + 1].x - in[2 * q].x).is_zero()) return;  // flagged by both
  const Affine<Fq28> a = aff[q];
  const XYZZ<Fq28> x = xy[q];
  if (!(a.x * x.zz - x.x).is_zero() || !(a.y * x.zzz - x.y).is_zero()) atomicAdd(bad, 1u);
}

template <class L>
static double best_ms(int repeats, L launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  double best = 1e30;
  for (int r = 0; r < repeats; r++) {
    CK(hipEventRecord(a, nullptr));
    launch();
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms);
  }
  return best;
}

template <int NPL, int W>
static void run_affine(const Affine<Fq28>* in, Affine<Fq28>* aff, const XYZZ<Fq28>* xy, Fq28* scratch, uint32_t* d_cnt, uint64_t pairs, int repeats,
                       double xyzz_ms) {
  const unsigned waves = (unsigned)(pairs / (64 * NPL));
  CK(hipMemset(d_cnt, 0, 8));
  const double ms = best_ms(repeats, [&]() { hipLaunchKernelGGL((k_pair_affine<NPL, W>), dim3(waves), dim3(64), 0, nullptr, in, aff, scratch, d_cnt); });
  CK(hipGetLastError());
  CK(hipMemset(d_cnt, 0, 8));
  hipLaunchKernelGGL((k_pair_affine<NPL, W>), dim3(waves), dim3(64), 0, nullptr, in, aff, scratch, d_cnt);
  hipLaunchKernelGGL(k_compare, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, nullptr, in, aff, xy, pairs, d_cnt + 1);
  uint32_t cnt[2];
  CK(hipMemcpy(cnt, d_cnt, 8, hipMemcpyDeviceToHost));
  printf("affine  NPL=%-3d W=%d  %8.3f ms  %7.3f ns/add  %6.2f G add/s  vs xyzz x%.3f  flagged %u  mismatches %u\n", NPL, W, ms, ms * 1e6 / pairs,
         pairs / ms / 1e6, xyzz_ms / ms, cnt[0], cnt[1]);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 22;
  const int repeats = argc > 2 ? atoi(argv[2]) : 5;
  const uint64_t pairs = 1ull << lg;
  Affine<Fq28>*in, *aff;
  XYZZ<Fq28>* xy;
  Fq28* scratch;
  uint32_t* d_cnt;
  CK(hipMalloc(&in, sizeof(Affine<Fq28>) * 2 * pairs));
  CK(hipMalloc(&aff, sizeof(Affine<Fq28>) * pairs));
  CK(hipMalloc(&xy, sizeof(XYZZ<Fq28>) * pairs));
  CK(hipMalloc(&scratch, sizeof(Fq28) * pairs));
  CK(hipMalloc(&d_cnt, 8));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((2 * pairs + 255) / 256)), dim3(256), 0, nullptr, in, 2 * pairs, 0x5eedull, 1u << 16);
  CK(hipDeviceSynchronize());
  printf("pairs 2^%d (%.1f MB of points in, %.1f MB of prefix products), best of %d launches\n", lg, 2.0 * pairs * sizeof(Affine<Fq28>) / 1e6,
         (double)pairs * sizeof(Fq28) / 1e6, repeats);
  CK(hipMemset(d_cnt, 0, 8));
  const double x3 = best_ms(repeats, [&]() { hipLaunchKernelGGL((k_pair_xyzz<3>), dim3((unsigned)(pairs / 64)), dim3(64), 0, nullptr, in, xy, d_cnt); });
  CK(hipGetLastError());
  printf("xyzz    W=3          %8.3f ms  %7.3f ns/add  %6.2f G add/s\n", x3, x3 * 1e6 / pairs, pairs / x3 / 1e6);
  run_affine<8, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<16, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<32, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<64, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<128, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<256, 3>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  run_affine<32, 2>(in, aff, xy, scratch, d_cnt, pairs, repeats, x3);
  return 0;
}

// Field-level experiments behind two design decisions of the G1 bucket accumulation:
//
//  A. 381-bit Montgomery product on the fp64 FMA pipe (8 limbs of 48 bits in doubles, the
//     Emmart-Zheng-Weems split: hi = fma_rz(a, b, 2^100), lo = fma_rz(a, b, -hi'), integer accumulation of
//     the bit patterns) against the product the kernels use (14 signed 28-bit limbs, v_mad_i64_i32 into
//     64-bit columns).  Both run as a dependent chain x <- x * y in registers, 2 waves per SIMD.
//  B. Batched-affine bucket additions (Montgomery's trick: one field inversion shared by k independent
//     affine additions of one lane, operands and prefix products in HBM) against the XYZZ mixed
//     addition with the accumulator in registers, for k = 16 .. 1024.
//
// Prints s_memtime ticks per operation per wave and wall-clock rates; run under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ... -- scripts/_bin/ubench_field
// for the instruction counts quoted in DESIGN.md.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-apps_amd/csrc scripts/ubench_field.hip -o scripts/_bin/ubench_field
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "curve.hpp"
#include "field28.hpp"

using namespace zkmi;

// ---------------------------------------------------------------------------------------------
// A. fp64 product.  Limbs: 8 x 48 bits (384 bits, R = 2^384), values kept as exact integers in doubles.
// a_i b_j < 2^96.  With round-toward-zero:
//   ph = fma(a, b, 2^100 * 1.5)      -> 1.5 * 2^100 + (a b rounded down to a multiple of 2^48)   [ulp at 2^100 is 2^48]
//   pl = fma(a, b, 1.5 * 2^100 - ph) -> a b mod 2^48   (exact: the difference is < 2^48)
// The integer value of ph's mantissa field is (1.5 * 2^100 >> 48) + floor(a b / 2^48): summing the raw bit
// patterns as int64 and subtracting the constant once per column gives the column sums without any
// floating-point additions of the products.
// ---------------------------------------------------------------------------------------------
struct Fq48 {
  double l[8];
};
__constant__ double P48[8];       // modulus limbs as doubles
__constant__ double PINV48;       // -p^-1 mod 2^48 as a double

__device__ __forceinline__ int64_t bits(double x) { return __double_as_longlong(x); }

// raw accumulation: *hi += bit pattern of (C + floor(a b / 2^48) 2^48), *lo += bit pattern of (2^52 + a b mod 2^48);
// the constants are taken off once per column (the number of terms per column is known at compile time)
__device__ __forceinline__ void mac48(int64_t* lo, int64_t* hi, double a, double b) {
  const double C = 0x1.8p100, C2 = 0x1.8p100 + 0x1p52;
  const double ph = __builtin_fma(a, b, C);        // round-toward-zero mode is set for the whole kernel
  const double pl = __builtin_fma(a, b, C2 - ph);  // exact: a b - H + 2^52
  *hi += bits(ph);
  *lo += bits(pl);
}
__device__ __forceinline__ double u48_to_double(int64_t t) {
  // 48-bit integer -> double without a 64-bit convert: two 24-bit halves
  const uint32_t lo24 = (uint32_t)t & 0xffffffu, hi24 = (uint32_t)(t >> 24) & 0xffffffu;
  return __builtin_fma((double)hi24, 0x1p24, (double)lo24);
}

__device__ __forceinline__ Fq48 mul48(const Fq48& a, const Fq48& b) {
  int64_t T[17];
#pragma unroll
  for (int i = 0; i < 17; i++) T[i] = 0;
  const int64_t HB = bits(0x1.8p100), LB = bits(0x1p52);
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) mac48(&T[i + j], &T[i + j + 1], a.l[i], b.l[j]);
  // column c received min(c, 14 - c) + 1 low parts (c <= 14) and as many high parts as column c - 1 has low parts
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int nlo = c <= 14 ? ((c < 8 ? c : 14 - c) + 1) : 0;
    const int nhi = c >= 1 ? (((c - 1) < 8 ? (c - 1) : 14 - (c - 1)) + 1) : 0;
    T[c] -= nlo * LB + nhi * HB;
  }
  // Montgomery reduction, one limb per round
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const double td = u48_to_double(T[k] & ((1ll << 48) - 1));
    const double ph = __builtin_fma(td, PINV48, 0x1.8p100);
    const double m = __builtin_fma(td, PINV48, 0x1.8p100 - ph);  // t * pinv mod 2^48
    int64_t lo[9];
#pragma unroll
    for (int j = 0; j < 9; j++) lo[j] = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) mac48(&lo[j], &lo[j + 1], m, P48[j]);
#pragma unroll
    for (int j = 0; j < 9; j++) T[k + j] += lo[j] - ((j < 8 ? LB : 0) + (j > 0 ? HB : 0));
    T[k + 1] += T[k] >> 48;
  }
  Fq48 r;
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int64_t v = T[8 + k] + c;
    r.l[k] = u48_to_double(v & ((1ll << 48) - 1));
    c = v >> 48;
  }
  return r;
}

template <int WHICH>
__global__ void __launch_bounds__(256, 2) k_modmul(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  uint64_t acc = 0;
  uint64_t t0, t1;
  // double-precision rounding mode = toward zero for the fp64 variant: MODE[3:2] = 3 (hwreg id 1, offset 2, size 2)
  if (WHICH == 1) __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
  if (WHICH == 0) {
    Fq28 x, y;
    for (int i = 0; i < 14; i++) {
      x.l[i] = (int32_t)((seed * 2654435761u + threadIdx.x * 40503u + i * 977u) & 0xfffffff);
      y.l[i] = (int32_t)((seed * 40503u + threadIdx.x * 2654435761u + i * 131u) & 0xfffffff);
    }
    x.l[13] &= 0xffff;
    y.l[13] &= 0xffff;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) x = Fq28::mul_inline(x, y);
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 14; i++) acc += (uint32_t)x.l[i];
  } else {
    Fq48 x, y;
    for (int i = 0; i < 8; i++) {
      x.l[i] = (double)(((uint64_t)(seed * 2654435761u + threadIdx.x * 40503u + i * 977u) << 16) & ((1ull << 48) - 1));
      y.l[i] = (double)(((uint64_t)(seed * 40503u + threadIdx.x * 2654435761u + i * 131u) << 16) & ((1ull << 48) - 1));
    }
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) x = mul48(x, y);
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; i++) acc += (uint64_t)x.l[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------
// B. batched-affine additions.  Lane t owns k pairs (P_j, Q_j) of distinct affine points
// (pts[2 (t k + j)], pts[2 (t k + j) + 1]) and produces R_j = P_j + Q_j (affine) in place of P_j:
//   forward : d_j = x(Q_j) - x(P_j);  pre[j] = d_0 ... d_{j-1}   (stored in HBM: 56 B per pair)
//   one Fermat inversion of the running product
//   backward: 1/d_j = inv * pre[j];  inv *= d_j;  lambda = (y_Q - y_P)/d_j;  x3 = lambda^2 - x_P - x_Q;
//             y3 = lambda (x_P - x3) - y_P
// 5 M + 1 S per addition + (one inversion)/k, against 7 M + 2 S + one double product for the XYZZ mixed addition.
// ---------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T ldg(const T* p) {
  T r;
  const uint4* s = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
  return r;
}
template <class T>
__device__ __forceinline__ void stg(T* p, const T& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
}
struct alignas(16) Fq28Pad {  // 56-byte element padded to 64 for 16-byte vector access
  Fq28 v;
  int32_t pad[2];
};

__global__ void __launch_bounds__(256, 2)
k_batched_affine(Affine<Fq28>* __restrict__ pts, Fq28Pad* __restrict__ pre, uint32_t lanes, uint32_t k, uint64_t* cyc) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  // lane-interleaved layout: pair j of lane t lives at index j * lanes + t (coalesced across the wave)
  Fq28 run = Fq28::one();
  for (uint32_t j = 0; j < k; j++) {
    const size_t idx = (size_t)j * lanes + t;
    const Affine<Fq28> p = ldg(pts + 2 * idx), q = ldg(pts + 2 * idx + 1);
    Fq28Pad w;
    w.v = run;
    w.pad[0] = w.pad[1] = 0;
    stg(pre + idx, w);
    run = run * (q.x - p.x);
  }
  Fq28 inv = run.inv();
  for (uint32_t j = k; j-- > 0;) {
    const size_t idx = (size_t)j * lanes + t;
    const Affine<Fq28> p = ldg(pts + 2 * idx), q = ldg(pts + 2 * idx + 1);
    const Fq28 d = q.x - p.x;
    const Fq28 dinv = inv * ldg(pre + idx).v;
    inv = inv * d;
    const Fq28 lam = f_sub_lazy(q.y, p.y) * dinv;
    Affine<Fq28> r;
    r.x = lam.sqr() - p.x - q.x;
    r.y = f_mul_sub_mul(lam, f_sub_lazy(p.x, r.x), p.y, Fq28::one());
    stg(pts + 2 * idx, r);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

// the same 2 k points folded into ONE XYZZ accumulator per lane by mixed additions (what k_accum does)
__global__ void __launch_bounds__(256, 2)
k_xyzz_chain(const Affine<Fq28>* __restrict__ pts, XYZZ<Fq28>* __restrict__ out, uint32_t lanes, uint32_t k, uint64_t* cyc) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  XYZZ<Fq28> acc = XYZZ<Fq28>::infinity();
  for (uint32_t j = 0; j < k; j++) {
    const size_t idx = (size_t)j * lanes + t;
    acc.madd(ldg(pts + 2 * idx + 1));  // k mixed additions, k gathers (the affine kernel reads 2 k points)
  }
  stg(out + t, acc);
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

// distinct points on the curve: P_i = (i + 1) G by repeated addition inside each lane (setup only)
__global__ void k_fill_points(Affine<Fq28>* pts, uint32_t n, Affine<Fq28> g) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  XYZZ<Fq28> acc = XYZZ<Fq28>::from_affine(g);
  uint32_t e = t + 2;  // (t + 2) G by double-and-add
  XYZZ<Fq28> res = XYZZ<Fq28>::infinity();
  while (e) {
    if (e & 1) res.add(acc);
    acc.dbl_inplace();
    e >>= 1;
  }
  stg(pts + t, res.to_affine());
}

static double median_ticks(uint64_t* d_cyc, size_t waves) {
  std::vector<uint64_t> h(waves);
  hipMemcpy(h.data(), d_cyc, 8 * waves, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[waves / 2];
}

int main() {
  // --- constants of the fp64 form
  {
    // p in 48-bit limbs and -p^-1 mod 2^48, from the 28-bit constants
    unsigned __int128 acc = 0;
    int accb = 0, k = 0;
    double p48[8];
    for (int i = 0; i < 14; i++) {
      acc |= (unsigned __int128)(uint32_t)Fq28Params::MOD[i] << accb;
      accb += 28;
      while (accb >= 48 && k < 8) {
        p48[k++] = (double)(uint64_t)(acc & ((1ull << 48) - 1));
        acc >>= 48;
        accb -= 48;
      }
    }
    while (k < 8) { p48[k++] = (double)(uint64_t)(acc & ((1ull << 48) - 1)); acc >>= 48; }
    const uint64_t p0 = (uint64_t)p48[0];
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;  // p0^-1 mod 2^64
    const double pinv = (double)((0 - inv) & ((1ull << 48) - 1));
    hipMemcpyToSymbol(HIP_SYMBOL(P48), p48, sizeof(p48));
    hipMemcpyToSymbol(HIP_SYMBOL(PINV48), &pinv, sizeof(pinv));
  }
  const int cus = 256;
  uint64_t *d_out, *d_cyc;
  hipMalloc(&d_out, 8ull * cus * 512);
  hipMalloc(&d_cyc, 8ull * cus * 8 * 64);
  printf("A. dependent chain of 381-bit Montgomery products, 2 waves per SIMD (512 threads per CU), %d CUs\n", cus);
  for (int which = 0; which < 2; which++) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_modmul<0>, dim3(2 * cus), dim3(256), 0, 0, d_out, d_cyc, iters, 7u);
      else hipLaunchKernelGGL(k_modmul<1>, dim3(2 * cus), dim3(256), 0, 0, d_out, d_cyc, iters, 7u);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ticks = median_ticks(d_cyc, 2 * cus * 4) / iters;
    const double total = (double)iters * 2 * cus * 256;
    printf("  %-34s %8.1f ticks per product per wave  %7.2f G products/s chip-wide  (%.3f ms)\n",
           which == 0 ? "14 x 28-bit limbs, v_mad_i64_i32" : "8 x 48-bit limbs, v_fma_f64 split", ticks, total / (ms * 1e-3) / 1e9, ms);
  }

  printf("B. bucket additions: batched affine (one inversion per k additions per lane) vs XYZZ mixed additions\n");
  const uint32_t lanes = 2 * cus * 256;  // 2 waves per SIMD
  const uint32_t kmax = 1024;
  Affine<Fq28>* d_pts;
  Fq28Pad* d_pre;
  XYZZ<Fq28>* d_acc;
  hipMalloc(&d_pts, sizeof(Affine<Fq28>) * 2ull * lanes * kmax);  // 30 GB at kmax = 1024
  hipMalloc(&d_pre, sizeof(Fq28Pad) * (size_t)lanes * kmax);
  hipMalloc(&d_acc, sizeof(XYZZ<Fq28>) * lanes);
  Affine<Fq28> g;
  {
    // generator in the 28-bit representation, through the host conversions of the library headers
    Fq gx, gy;
    static const uint32_t GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                    0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
    static const uint32_t GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                    0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    for (int i = 0; i < 12; i++) { gx.l[i] = GX[i]; gy.l[i] = GY[i]; }
    g = {Fq28::from_canonical(gx.l), Fq28::from_canonical(gy.l)};
  }
  const size_t npts = 2ull * lanes * kmax;
  // fill with a repeating pattern of 2^16 distinct points (setup cost; pairs at distance lanes are distinct)
  const uint32_t distinct = 1u << 16;
  Affine<Fq28>* d_small;
  hipMalloc(&d_small, sizeof(Affine<Fq28>) * distinct);
  hipLaunchKernelGGL(k_fill_points, dim3(distinct / 64), dim3(64), 0, 0, d_small, distinct, g);
  hipDeviceSynchronize();
  for (size_t off = 0; off < npts; off += distinct - 1)  // stride distinct-1: neighbours in a pair differ
    hipMemcpy(d_pts + off, d_small, sizeof(Affine<Fq28>) * std::min<size_t>(distinct - 1, npts - off), hipMemcpyDeviceToDevice);
  hipDeviceSynchronize();
  // C. does the dispatcher keep 2 big-register waves per SIMD resident when it has to REFILL slots?  Same mixed-addition
  // loop, k = 32 additions per lane, grid = 1x / 4x / 8x the resident capacity (2048 waves), workgroups of 256 and 64
  // threads: with perfect refill the time scales with the number of rounds.
  printf("C. refill: k_xyzz_chain, 32 additions per lane, grids of R x 2048 waves\n");
  for (uint32_t bs : {256u, 64u}) {
    for (uint32_t rounds : {1u, 4u, 8u}) {
      const uint32_t L = lanes * rounds, k = 32;
      if ((size_t)L * k * 2 > npts) continue;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      XYZZ<Fq28>* d_acc2;
      hipMalloc(&d_acc2, sizeof(XYZZ<Fq28>) * L);
      uint64_t* d_cyc2;
      hipMalloc(&d_cyc2, 8ull * (L / 64));
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_xyzz_chain, dim3(L / bs), dim3(bs), 0, 0, d_pts, d_acc2, L, k, d_cyc2);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("  block %3u, %u round(s): %7.3f ms  (%.3f ms per round)  %6.2f G adds/s\n", bs, rounds, ms, ms / rounds, (double)L * k / (ms * 1e-3) / 1e9);
      hipFree(d_acc2);
      hipFree(d_cyc2);
    }
  }
  if (getenv("UBENCH_SKIP_B")) return 0;
  for (uint32_t k : {16u, 64u, 256u, 1024u}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms_aff = 0, ms_x = 0;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_xyzz_chain, dim3(lanes / 256), dim3(256), 0, 0, d_pts, d_acc, lanes, k, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms_x, e0, e1);
    const double tx = median_ticks(d_cyc, lanes / 64) / k;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_batched_affine, dim3(lanes / 256), dim3(256), 0, 0, d_pts, d_pre, lanes, k, d_cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms_aff, e0, e1);
    const double ta = median_ticks(d_cyc, lanes / 64) / k;
    const double adds = (double)lanes * k;
    printf("  k = %4u: XYZZ mixed add %8.0f ticks/add/wave %7.2f G adds/s (%.2f ms) | batched affine %8.0f ticks/add/wave %7.2f G adds/s (%.2f ms)"
           "  -> affine/XYZZ time = %.2f\n", k, tx, adds / (ms_x * 1e-3) / 1e9, ms_x, ta, adds / (ms_aff * 1e-3) / 1e9, ms_aff, ms_aff / ms_x);
  }
  return 0;
}

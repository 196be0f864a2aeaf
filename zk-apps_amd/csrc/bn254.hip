// zkmi — BN254 (alt_bn128) instantiation of the MSM and NTT kernels and a KZG-commit-shaped
// driver.
//
// BN254 is the curve of the reference's actual proving stack: its relations are halo2 circuits over
// halo2curves::bn256 (shielder/Cargo.toml:26, shielder/Cargo.lock:436-492); a halo2/KZG prover
// spends its time in exactly two primitives, which the reference reaches only through crates that
// are not in the tree:
//   halo2_proofs::arithmetic::best_fft        -> radix-2 NTT over bn256::Fr (two-adicity 28,
//                                                ROOT_OF_UNITY = 7^((r-1)/2^28))
//   halo2_proofs::arithmetic::best_multiexp   -> G1 MSM (y^2 = x^3 + 3, generator (1, 2))
//   ParamsKZG::commit_lagrange                -> commitment = MSM(SRS in Lagrange basis, evaluations);
//                                                commit(coefficients) = MSM(SRS, iNTT(evaluations))
//   halo2_proofs::arithmetic::eval_polynomial -> p(zeta) (Horner)
//   halo2_proofs::arithmetic::kate_division   -> q(X) = (p(X) - p(zeta)) / (X - zeta) by synthetic division; the KZG
//                                                opening proof is commit(q) (poly::kzg::multiopen, one polynomial, one point)
// The kernels are the same templates as the BLS12-381 ones (msm_impl.hpp over Fp28<BnFq28Params>,
// ntt.hip over Fp28<BnFr28Params>): 10 limbs instead of 14, and the bucket accumulation fits
// 166 VGPRs = 3 waves per SIMD instead of 2.
// Wire formats: Fr / Fq = 32-byte little-endian canonical integers; G1 affine = x || y (64 B),
// all-zero = the point at infinity.
#include <string.h>
#include <new>
#include <vector>
#include "ctx.hpp"

using namespace zkmi;

struct zkmi_bn_bases {
  zkmi_ctx* ctx;
  Affine<BnFq28>* d28 = nullptr;  // device MSM representation (10 limbs, R = 2^280)
  Affine<BnFq28>* tab = nullptr;  // optional: 2^(c w) * P_i for every digit position w (zkmi_bn254_srs_prepare)
  uint64_t n = 0;
};

namespace {

bool words_lt(const uint32_t w[8], const uint32_t mod[8]) {
  for (int i = 7; i >= 0; i--)
    if (w[i] != mod[i]) return w[i] < mod[i];
  return false;
}
bool all_zero(const uint8_t* b, size_t n) {
  uint8_t acc = 0;
  for (size_t i = 0; i < n; i++) acc |= b[i];
  return acc == 0;
}
BnFq28 small(uint32_t v) {
  uint32_t w[8] = {v, 0, 0, 0, 0, 0, 0, 0};
  return BnFq28::from_canonical(w);
}
bool bn_from_wire(const uint8_t* b, Affine<BnFq28>* out, bool check) {
  if (all_zero(b, 64)) {
    *out = Affine<BnFq28>::infinity();
    return true;
  }
  uint32_t x[8], y[8];
  memcpy(x, b, 32);
  memcpy(y, b + 32, 32);
  if (!words_lt(x, BnFq28Params::MOD32) || !words_lt(y, BnFq28Params::MOD32)) return false;
  out->x = BnFq28::from_canonical(x);
  out->y = BnFq28::from_canonical(y);
  if (!check) return true;
  const BnFq28 lhs = out->y.sqr(), rhs = out->x.sqr() * out->x + small(3);
  return (lhs - rhs).is_zero();  // difference of two products: within is_zero's exact range
}
void bn_to_wire(const Affine<BnFq28>& p, uint8_t* b) {
  uint32_t x[8], y[8];
  p.x.to_canonical(x);
  p.y.to_canonical(y);
  memcpy(b, x, 32);
  memcpy(b + 32, y, 32);  // infinity = (0, 0) -> all zero
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only
// P_i = [1 + i * 0xC0FFEE] G, G = (1, 2): distinct points manufactured in HBM for tests / timing
__global__ void __launch_bounds__(64) k_bn_synth(Affine<BnFq28>* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0}, two[8] = {2, 0, 0, 0, 0, 0, 0, 0};
  const Affine<BnFq28> g = {BnFq28::from_canonical(one), BnFq28::from_canonical(two)};
  const uint64_t s = 1ull + i * 0xC0FFEEull;
  const uint32_t k[2] = {(uint32_t)s, (uint32_t)(s >> 32)};
  const XYZZ<BnFq28> r = scalar_mul(XYZZ<BnFq28>::from_affine(g), k, 2);
  const Affine<BnFq28> a = r.to_affine();
  uint4* d = reinterpret_cast<uint4*>(out + i);
  const uint4* sgm = reinterpret_cast<const uint4*>(&a);
#pragma unroll
  for (unsigned q = 0; q < sizeof(a) / 16; q++) d[q] = sgm[q];
}
#endif  // ZKMI_TESTING

// ---- KZG opening: eval_polynomial + kate_division as ONE suffix scan ----------------------------------------------
// With y = zeta, the synthetic division by (X - y) is q_i = sum_{j > i} p_j y^(j-i-1), and p(y) = sum_j p_j y^j: both are
// values of the inclusive suffix scan  S_i = sum_{j >= i} p_j y^(j-i)  (q_i = S_{i+1}, p(y) = S_0) -- a linear recurrence
// S_i = p_i + y S_{i+1} walked from the top coefficient down, which CPUs run as one dependent chain of n products.  Here:
//   level 1: 256-thread blocks own 1024 consecutive coefficients; a thread runs the recurrence over its four, the 256
//            thread values are combined by a log-step suffix scan in LDS (multipliers y^(4 2^s));
//   level 2: the block values V_k = S restricted to block k are the coefficients of the same problem in Y = y^1024
//            (1024 blocks per 2^20 coefficients), level 3 the same again in y^(2^20) (at most 64 values for 2^26);
//   then the levels are walked back down: a block repeats its scan with the suffix value at the start of the NEXT block
//   as an extra, 257th entry, and writes S for all of its elements.
// Every multiplier is a power zeta^(2^j): one table of 30 entries (k_kzg_pow_table) serves all levels.
constexpr int KZG_T = 256, KZG_E = 4, KZG_L = KZG_T * KZG_E;  // threads per block, elements per thread, elements per block
constexpr int KZG_POWS = 32;

__global__ void k_kzg_pow_table(BnFr28 zeta, BnFr28* __restrict__ pows) {
  if (threadIdx.x || blockIdx.x) return;
  BnFr28 v = zeta;
  for (int j = 0; j < KZG_POWS; j++) {
    pows[j] = v;  // zeta^(2^j)
    v = v.sqr();
  }
}

// in: m elements -- canonical words (CANON_IN, level 1) or limb form (the block values of the level below).
// FINAL = false: vals[blockIdx.x] = the block's own suffix value at its first element (no carry).
// FINAL = true:  carry[blockIdx.x + 1] (0 behind the last block) joins as the 257th entry and S_i is written for every element:
//                level 1 (CANON_OUT): q[i - 1] = S_i for i >= 1 in canonical words and *eval = S_0; otherwise out[i] = S_i in limb form.
// lg = log2 of the exponent of zeta this level's variable is (0, 10, 20).
template <bool CANON_IN, bool FINAL, bool CANON_OUT>
__global__ void __launch_bounds__(KZG_T)
k_kzg_scan(const void* __restrict__ in, uint64_t m, const BnFr28* __restrict__ pows, int lg, const BnFr28* __restrict__ carry, uint32_t n_carry,
           BnFr28* __restrict__ vals, void* __restrict__ out, uint32_t* __restrict__ eval_out, const void* const* __restrict__ in_table = nullptr,
           uint64_t in_stride = 0, uint64_t vals_stride = 0) {
  __shared__ BnFr28 sh[KZG_T + 1];
  const int t = threadIdx.x;
  // blockIdx.y = polynomial of a set (block values only: zkmi_bn254_kzg_open_many_dev evaluates k polynomials per launch):
  // level 1 reads polynomial y through a table of pointers, the levels above read row y of the level below
  if (in_table) in = in_table[blockIdx.y];
  else if (in_stride) in = static_cast<const BnFr28*>(in) + (size_t)blockIdx.y * in_stride;
  if (vals) vals += (size_t)blockIdx.y * vals_stride;
  const uint64_t first = (uint64_t)blockIdx.x * KZG_L + (uint64_t)t * KZG_E;
  BnFr28 c[KZG_E];
#pragma unroll
  for (int e = 0; e < KZG_E; e++) {
    const uint64_t i = first + e;
    if (i >= m) {
      c[e] = BnFr28::zero();
    } else if constexpr (CANON_IN) {
      const uint4* w4 = reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(in) + 8 * i);
      const uint4 a = w4[0], b = w4[1];
      const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      c[e] = BnFr28::from_canonical(w);
    } else {
      c[e] = static_cast<const BnFr28*>(in)[i];
    }
  }
  const BnFr28 y = pows[lg];
  BnFr28 v = c[KZG_E - 1];
#pragma unroll
  for (int e = KZG_E - 2; e >= 0; e--) v = c[e] + y * v;  // the thread's own suffix value at its first element
  sh[t] = v;
  if (t == 0) {
    BnFr28 cin = BnFr28::zero();
    if (FINAL && carry && blockIdx.x + 1 < n_carry) cin = carry[blockIdx.x + 1];
    sh[KZG_T] = cin;
  }
  __syncthreads();
  // suffix scan over the 257 entries: entry t covers [t, t + 2^s) after step s; the multiplier is y^(E 2^s)
  for (int s = 0; (1 << s) <= KZG_T; s++) {
    const int src = t + (1 << s);
    BnFr28 add = BnFr28::zero();
    const bool on = src <= KZG_T;
    if (on) add = pows[lg + 2 + s] * sh[src];
    __syncthreads();
    if (on) sh[t] = sh[t] + add;
    __syncthreads();
  }
  if constexpr (!FINAL) {
    if (t == 0) vals[blockIdx.x] = sh[0];
    return;
  } else {
    // S at the first element of the NEXT thread (entry 256 = the carry), then the recurrence over this thread's four
    BnFr28 nxt = sh[t + 1];
    // (sh[t + 1] after the scan covers [t + 1, 257): everything behind this thread, the carry included)
#pragma unroll
    for (int e = KZG_E - 1; e >= 0; e--) {
      const uint64_t i = first + e;
      nxt = c[e] + y * nxt;  // S_i
      if (i >= m) continue;
      if constexpr (CANON_OUT) {
        uint32_t w[8];
        nxt.to_canonical(w);
        if (i == 0) {
#pragma unroll
          for (int k = 0; k < 8; k++) eval_out[k] = w[k];
        } else {
          uint4* d = reinterpret_cast<uint4*>(static_cast<uint32_t*>(out) + 8 * (i - 1));
          d[0] = make_uint4(w[0], w[1], w[2], w[3]);
          d[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
      } else {
        static_cast<BnFr28*>(out)[i] = nxt;
      }
    }
  }
}

// f_i = sum_j v^j p_j[i] (Horner in v from the last polynomial down), canonical words in and out: the polynomial whose
// quotient opens all k at once
__global__ void __launch_bounds__(256)
k_kzg_lincomb(const void* const* __restrict__ polys, uint32_t k, uint64_t n, BnFr28 v, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  BnFr28 acc = BnFr28::zero();
  for (int j = (int)k - 1; j >= 0; j--) {
    const uint4* w4 = reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(polys[j]) + 8 * i);
    const uint4 a = w4[0], b = w4[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    acc = acc * v + BnFr28::from_canonical(w);
  }
  uint32_t w[8];
  acc.to_canonical(w);
  uint4* d = reinterpret_cast<uint4*>(out + 8 * i);
  d[0] = make_uint4(w[0], w[1], w[2], w[3]);
  d[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// tops[j * stride] (limb form) -> out[8 j ..] canonical
__global__ void k_kzg_evals_out(const BnFr28* __restrict__ tops, uint64_t stride, uint32_t k, uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  uint32_t w[8];
  tops[(size_t)j * stride].to_canonical(w);
#pragma unroll
  for (int q = 0; q < 8; q++) out[8 * j + q] = w[q];
}

// ---- permutation-argument grand product: z_0 = 1, z_(i+1) = z_i num_i / den_i ---------------------------------------
// (halo2_proofs::plonk::permutation::prover::commit: batch_invert of the denominators, then a running product -- one
// dependent chain of n products on a CPU.)  k_gp_ratio: a thread inverts its eight denominators with Montgomery's trick and
// one Fermat inversion of its own (47 products per element; the MSMs of a prover cost thousands per element), ratio_i =
// num_i / den_i in limb form.  k_gp_scan: the EXCLUSIVE prefix products of the ratios with the three-level shape of
// k_kzg_scan (1024 elements per block, block totals one level up, carries on the way back down).
constexpr int GP_E = 8;
__device__ __forceinline__ BnFr28 ld_canon_fr(const void* base, uint64_t i) {
  const uint4* w4 = reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(base) + 8 * i);
  const uint4 a = w4[0], b = w4[1];
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return BnFr28::from_canonical(w);
}
__global__ void __launch_bounds__(64)
k_gp_ratio(const void* __restrict__ num, const void* __restrict__ den, uint64_t n, BnFr28* __restrict__ ratio, uint32_t* __restrict__ zero_flag) {
  const uint64_t first = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * GP_E;
  if (first >= n) return;
  BnFr28 pre[GP_E];  // pre[e] = den_0 ... den_e (entries beyond n count as 1)
  BnFr28 acc = BnFr28::one();
  bool zero = false;
#pragma unroll
  for (int e = 0; e < GP_E; e++) {
    if (first + e < n) {
      const uint4* w4 = reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(den) + 8 * (first + e));
      const uint4 a = w4[0], b = w4[1];
      zero |= (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0;  // canonical input: zero is all-zero words
      acc = acc * ld_canon_fr(den, first + e);
    }
    pre[e] = acc;
  }
  if (zero) {
    atomicOr(zero_flag, 1u);
    return;
  }
  BnFr28 inv = acc.inv();  // 1 / (den_0 ... den_last)
  for (int e = GP_E - 1; e >= 0; e--) {
    if (first + e < n) {
      const BnFr28 inv_e = e ? inv * pre[e ? e - 1 : 0] : inv;  // 1 / den_e
      if (e) inv = inv * ld_canon_fr(den, first + e);
      ratio[first + e] = ld_canon_fr(num, first + e) * inv_e;
    }
  }
}

// FINAL = false: vals[blockIdx.x] = product of the block's elements.
// FINAL = true:  carry[blockIdx.x] (1 when carry is null) = product of everything in front of the block; the exclusive prefix
//                products are written for every element -- canonical words (CANON_OUT, level 1; the product of ALL elements also
//                goes to total_out) or limb form.
template <bool FINAL, bool CANON_OUT>
__global__ void __launch_bounds__(KZG_T)
k_gp_scan(const BnFr28* __restrict__ in, uint64_t m, const BnFr28* __restrict__ carry, BnFr28* __restrict__ vals, void* __restrict__ out,
          uint32_t* __restrict__ total_out) {
  __shared__ BnFr28 sh[KZG_T];
  const int t = threadIdx.x;
  const uint64_t first = (uint64_t)blockIdx.x * KZG_L + (uint64_t)t * KZG_E;
  BnFr28 c[KZG_E];
  BnFr28 v = BnFr28::one();
#pragma unroll
  for (int e = 0; e < KZG_E; e++) {
    c[e] = first + e < m ? in[first + e] : BnFr28::one();
    v = v * c[e];
  }
  sh[t] = v;
  __syncthreads();
  for (int s = 1; s < KZG_T; s <<= 1) {  // inclusive prefix products over the thread values
    BnFr28 left = BnFr28::one();
    const bool on = t >= s;
    if (on) left = sh[t - s];
    __syncthreads();
    if (on) sh[t] = sh[t] * left;
    __syncthreads();
  }
  if constexpr (!FINAL) {
    if (t == KZG_T - 1) vals[blockIdx.x] = sh[t];
    return;
  } else {
    BnFr28 run = carry ? carry[blockIdx.x] : BnFr28::one();
    if (t > 0) run = run * sh[t - 1];  // everything in front of this thread's first element
#pragma unroll
    for (int e = 0; e < KZG_E; e++) {
      const uint64_t i = first + e;
      if (i >= m) break;
      if constexpr (CANON_OUT) {
        uint32_t w[8];
        run.to_canonical(w);
        uint4* d = reinterpret_cast<uint4*>(static_cast<uint32_t*>(out) + 8 * i);
        d[0] = make_uint4(w[0], w[1], w[2], w[3]);
        d[1] = make_uint4(w[4], w[5], w[6], w[7]);
      } else {
        static_cast<BnFr28*>(out)[i] = run;
      }
      run = run * c[e];
      if (CANON_OUT && i == m - 1) {
        uint32_t w[8];
        run.to_canonical(w);
#pragma unroll
        for (int k = 0; k < 8; k++) total_out[k] = w[k];
      }
    }
  }
}

bool scalars_canonical(const uint8_t* s, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    uint32_t w[8];
    memcpy(w, s + 32 * i, 32);
    if (!words_lt(w, BnFr28Params::MOD32)) return false;
  }
  return true;
}

hipError_t work_buffer(zkmi_ctx* ctx, uint64_t bytes) {
  if (ctx->d_work_cap >= bytes) return hipSuccess;
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  ctx->d_work = nullptr;
  ctx->d_work_cap = 0;
  hipError_t e = hipMalloc(&ctx->d_work, bytes);
  if (e == hipSuccess) ctx->d_work_cap = bytes;
  return e;
}

int32_t msm_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bn_bases* bases, uint8_t out[64]) {
  // a prepared SRS (fixed bases, full length): every digit of a scalar goes to ONE set of buckets through
  // the table of 2^(c w) multiples -- 13 insertions per scalar instead of 16, 16 partition sums instead of
  // a Horner walk over the windows (the prover's shared-bucket schedule)
  const bool shared = bases->tab != nullptr && n == bases->n && n > 0;
  ZK_HIP(ctx, ctx->sort.reserve(n, shared));
  ZK_HIP(ctx, ctx->g1_bn.reserve(n, shared));
  if (shared)
    ZK_HIP(ctx, ctx->sort.run_shared(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  else
    ZK_HIP(ctx, ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  ZK_HIP(ctx, ctx->g1_bn.run_device(ctx->sort, shared ? bases->tab : bases->d28, ctx->stream, ctx->stream_aux,
                                    ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  XYZZ<BnFq> res;
  ZK_HIP(ctx, ctx->g1_bn.finish_host(&res));
  const Affine<BnFq> a = res.to_affine();
  const Affine<BnFq28> a28 = {fq28_from_fq(a.x), fq28_from_fq(a.y)};
  bn_to_wire(a28, out);
  return ZKMI_OK;
}

}  // namespace

uint64_t zkmi_layout_bn_bases() { return sizeof(zkmi_bn_bases); }  // capi.hip zkmi_abi_layout_probe

extern "C" {

int32_t zkmi_bn254_bases_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bn_bases** out) {
  ZK_ENTER(ctx);
  if (!out || (n && !affine) || n >= (1ull << 31)) return ZKMI_ERR_BAD_ARG;
  *out = nullptr;
  std::vector<Affine<BnFq28>> h(n);
  for (uint64_t i = 0; i < n; i++)
    if (!bn_from_wire(affine + 64 * i, &h[i], check != 0))
      return ctx->fail(ZKMI_ERR_NON_CANONICAL, "bn254 base point not canonical / not on curve");
  zkmi_bn_bases* b = new (std::nothrow) zkmi_bn_bases();
  if (!b) return ZKMI_ERR_BAD_ARG;
  b->ctx = ctx;
  b->n = n;
  hipError_t e = hipMalloc(&b->d28, sizeof(Affine<BnFq28>) * (n ? n : 1));
  if (e == hipSuccess && n) e = hipMemcpy(b->d28, h.data(), sizeof(Affine<BnFq28>) * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (b->d28) (void)hipFree(b->d28);
    delete b;
    return ctx->hip_fail(e, "bn254 bases upload");
  }
  *out = b;
  return ZKMI_OK;
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_bn254_bases_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bn_bases** out) {
  ZK_ENTER(ctx);
  if (!out || n >= (1ull << 31)) return ZKMI_ERR_BAD_ARG;
  zkmi_bn_bases* b = new (std::nothrow) zkmi_bn_bases();
  if (!b) return ZKMI_ERR_BAD_ARG;
  b->ctx = ctx;
  b->n = n;
  hipError_t e = hipMalloc(&b->d28, sizeof(Affine<BnFq28>) * (n ? n : 1));
  if (e == hipSuccess && n) {
    hipLaunchKernelGGL(k_bn_synth, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, b->d28, n);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) {
    if (b->d28) (void)hipFree(b->d28);
    delete b;
    return ctx->hip_fail(e, "bn254 synthetic bases");
  }
  *out = b;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

int32_t zkmi_bn254_bases_read(zkmi_ctx* ctx, const zkmi_bn_bases* b, uint64_t first, uint64_t count, uint8_t* out) {
  ZK_ENTER(ctx);
  if (!b || !out || first + count > b->n) return ZKMI_ERR_BAD_ARG;
  std::vector<Affine<BnFq28>> h(count);
  if (count) ZK_HIP(ctx, hipMemcpy(h.data(), b->d28 + first, sizeof(Affine<BnFq28>) * count, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < count; i++) bn_to_wire(h[i], out + 64 * i);
  return ZKMI_OK;
}

// Fixed-base preparation of an SRS: table[w * n + i] = 2^(c w) * P_i (13 x n points at n = 2^20, 1.1 GB).
// MSMs / commitments of exactly n terms over these bases then run the shared-bucket schedule.
int32_t zkmi_bn254_srs_prepare(zkmi_ctx* ctx, zkmi_bn_bases* b) {
  ZK_ENTER(ctx);
  if (!b || b->n == 0) return ZKMI_ERR_BAD_ARG;
  if (b->tab) return ZKMI_OK;
  const MsmPlan plan = msm_make_plan_shared(b->n);
  // table indices (digit * n + point) share a 32-bit word with the sign bit
  if ((uint64_t)plan.ndigits * b->n >= (1ull << 31)) return ctx->fail(ZKMI_ERR_BAD_ARG, "SRS too long for a table");
  hipError_t e = msm_build_table<BnFq28>(b->d28, b->n, plan, &b->tab, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    if (b->tab) (void)hipFree(b->tab);
    b->tab = nullptr;
    return ctx->hip_fail(e, "bn254 srs table");
  }
  return ZKMI_OK;
}

int32_t zkmi_bn254_bases_free(zkmi_bn_bases* b) {
  if (!b) return ZKMI_ERR_BAD_ARG;
  if (b->d28) (void)hipFree(b->d28);
  if (b->tab) (void)hipFree(b->tab);
  delete b;
  return ZKMI_OK;
}

int32_t zkmi_bn254_msm_g1_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bn_bases* bases,
                              uint8_t out_affine[64]) {
  ZK_ENTER(ctx);
  if (!bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !d_scalars)) return ZKMI_ERR_BAD_ARG;
  return msm_dev(ctx, d_scalars, n, bases, out_affine);
}

int32_t zkmi_bn254_msm_g1(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bn_bases* bases,
                          uint8_t out_affine[64]) {
  ZK_ENTER(ctx);
  if (!bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !scalars)) return ZKMI_ERR_BAD_ARG;
  if (!scalars_canonical(scalars, n)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "bn254 scalar >= r");
  ZK_HIP(ctx, ctx->staging(n * 32 + 32));
  if (n) ZK_HIP(ctx, hipMemcpyAsync(ctx->d_tmp, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return msm_dev(ctx, ctx->d_tmp, n, bases, out_affine);
}

int32_t zkmi_bn254_ntt_fr_dev(zkmi_ctx* ctx, void* d_data, uint32_t log_n, int32_t inverse, int32_t coset) {
  ZK_ENTER(ctx);
  if (!d_data || log_n > 26) return ZKMI_ERR_BAD_ARG;
  hipError_t e;
  NttDomainBn* dom = ctx->domain_bn((int)log_n, &e);
  if (!dom) return ctx->hip_fail(e, "bn254 ntt domain init");
  const uint32_t n = 1u << log_n;
  ZK_HIP(ctx, work_buffer(ctx, (uint64_t)n * sizeof(BnFr28)));
  BnFr28* work = static_cast<BnFr28*>(ctx->d_work);
  ZK_HIP(ctx, ntt_from_canonical(static_cast<const uint32_t*>(d_data), work, n, ctx->stream));
  PhaseTimer* t = ctx->timer();
  if (t) t->begin(PH_NTT, ctx->stream);
  e = dom->transform(work, inverse != 0, coset != 0, ctx->stream);
  if (t) t->end(PH_NTT, ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "bn254 ntt transform");
  ZK_HIP(ctx, ntt_to_canonical(work, static_cast<uint32_t*>(d_data), n, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int32_t zkmi_bn254_ntt_fr(zkmi_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset) {
  ZK_ENTER(ctx);
  if (!data || log_n > 26) return ZKMI_ERR_BAD_ARG;
  const uint64_t n = 1ull << log_n;
  if (!scalars_canonical(data, n)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "bn254 ntt input >= r");
  ZK_HIP(ctx, ctx->staging(n * 32));
  ZK_HIP(ctx, hipMemcpyAsync(ctx->d_tmp, data, n * 32, hipMemcpyHostToDevice, ctx->stream));
  const int32_t rc = zkmi_bn254_ntt_fr_dev(ctx, ctx->d_tmp, log_n, inverse, coset);
  if (rc != ZKMI_OK) return rc;
  ZK_HIP(ctx, hipMemcpyAsync(data, ctx->d_tmp, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// Commitment to the polynomial given by its 2^log_n evaluations over the radix-2 domain, against an SRS
// in the monomial basis: coefficients = iNTT(evaluations) (left in d_evals, canonical form), commitment =
// MSM(srs, coefficients).  d_evals is overwritten with the coefficients.
int32_t zkmi_bn254_kzg_commit_dev(zkmi_ctx* ctx, void* d_evals, uint32_t log_n, const zkmi_bn_bases* srs,
                                  uint8_t out_commitment[64]) {
  ZK_ENTER(ctx);
  if (!d_evals || !srs || !out_commitment || log_n > 26 || srs->n < (1ull << log_n)) return ZKMI_ERR_BAD_ARG;
  const int32_t rc = zkmi_bn254_ntt_fr_dev(ctx, d_evals, log_n, 1, 0);
  if (rc != ZKMI_OK) return rc;
  return msm_dev(ctx, d_evals, 1ull << log_n, srs, out_commitment);
}

// KZG opening at zeta of the polynomial with n coefficients (canonical words in HBM, constant term first):
// *out_eval = p(zeta), out_proof = commit(q), q(X) = (p(X) - p(zeta)) / (X - zeta) -- n - 1 coefficients, also written to
// d_quotient (canonical) when that is not null.  srs holds at least n - 1 points [tau^i] G; a prepared SRS is used through
// its table (the quotient is zero-padded to the SRS's length).
static int32_t kzg_open_core(zkmi_ctx* ctx, const void* d_coeffs, uint64_t n, const uint32_t zw[8], const zkmi_bn_bases* srs, void* d_quotient,
                             uint8_t out_eval[32], uint8_t out_proof[64]) {
  const uint64_t nb1 = (n + KZG_L - 1) / KZG_L, nb2 = (nb1 + KZG_L - 1) / KZG_L;  // nb2 <= 64 for n < 2^28
  // the quotient as MSM scalars: n - 1 of them, zero-padded to the length of a prepared SRS
  const bool padded = srs->tab != nullptr && srs->n >= n - 1;
  const uint64_t qn = padded ? srs->n : n - 1;
  ZK_HIP(ctx, ctx->staging((qn ? qn : 1) * 32 + 32));
  uint32_t* d_q = static_cast<uint32_t*>(ctx->d_tmp);
  uint32_t* d_eval = d_q + 8 * (qn ? qn : 1);
  // work: pows | V1[nb1] | E1[nb1] | V2[nb2] | E2[nb2]
  ZK_HIP(ctx, work_buffer(ctx, sizeof(BnFr28) * (KZG_POWS + 2 * nb1 + 2 * nb2 + 4)));
  BnFr28* pows = static_cast<BnFr28*>(ctx->d_work);
  BnFr28 *v1 = pows + KZG_POWS, *e1 = v1 + nb1, *v2 = e1 + nb1, *e2 = v2 + nb2;
  const hipStream_t st = ctx->stream;
  PhaseTimer* t = ctx->timer();
  if (t) t->begin(PH_MISC, st);
  if (padded && qn > n - 1) ZK_HIP(ctx, hipMemsetAsync(d_q + 8 * (n - 1), 0, 32 * (qn - (n - 1)), st));
  hipLaunchKernelGGL(k_kzg_pow_table, dim3(1), dim3(64), 0, st, BnFr28::from_canonical(zw), pows);
  const BnFr28* carry1 = nullptr;
  if (nb1 > 1) {
    hipLaunchKernelGGL((k_kzg_scan<true, false, false>), dim3((unsigned)nb1), dim3(KZG_T), 0, st, d_coeffs, n, pows, 0, nullptr, 0u, v1, nullptr, nullptr);
    const BnFr28* carry2 = nullptr;
    if (nb2 > 1) {
      hipLaunchKernelGGL((k_kzg_scan<false, false, false>), dim3((unsigned)nb2), dim3(KZG_T), 0, st, v1, nb1, pows, 10, nullptr, 0u, v2, nullptr, nullptr);
      hipLaunchKernelGGL((k_kzg_scan<false, true, false>), dim3(1), dim3(KZG_T), 0, st, v2, nb2, pows, 20, nullptr, 0u, nullptr, e2, nullptr);
      carry2 = e2;
    }
    hipLaunchKernelGGL((k_kzg_scan<false, true, false>), dim3((unsigned)nb2), dim3(KZG_T), 0, st, v1, nb1, pows, 10, carry2, (uint32_t)nb2, nullptr, e1, nullptr);
    carry1 = e1;
  }
  hipLaunchKernelGGL((k_kzg_scan<true, true, true>), dim3((unsigned)nb1), dim3(KZG_T), 0, st, d_coeffs, n, pows, 0, carry1, (uint32_t)nb1, nullptr, d_q, d_eval);
  ZK_HIP(ctx, hipGetLastError());
  if (t) t->end(PH_MISC, st);
  ZK_HIP(ctx, hipMemcpyAsync(out_eval, d_eval, 32, hipMemcpyDeviceToHost, st));
  if (d_quotient && n > 1) ZK_HIP(ctx, hipMemcpyAsync(d_quotient, d_q, 32 * (n - 1), hipMemcpyDeviceToDevice, st));
  ZK_HIP(ctx, hipStreamSynchronize(st));
  if (n == 1) {  // a constant: the quotient is the zero polynomial, its commitment the point at infinity
    memset(out_proof, 0, 64);
    return ZKMI_OK;
  }
  return msm_dev(ctx, d_q, qn, srs, out_proof);
}

int32_t zkmi_bn254_kzg_open_dev(zkmi_ctx* ctx, const void* d_coeffs, uint64_t n, const uint8_t zeta[32], const zkmi_bn_bases* srs,
                                void* d_quotient, uint8_t out_eval[32], uint8_t out_proof[64]) {
  ZK_ENTER(ctx);
  if (!d_coeffs || !zeta || !srs || !out_eval || !out_proof || n == 0 || n > MSM_MAX_TERMS || srs->n + 1 < n) return ZKMI_ERR_BAD_ARG;
  if (!scalars_canonical(zeta, 1)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "bn254 zeta >= r");
  uint32_t zw[8];
  memcpy(zw, zeta, 32);
  return kzg_open_core(ctx, d_coeffs, n, zw, srs, d_quotient, out_eval, out_proof);
}

// k polynomials of n coefficients opened at ONE point (a rotation set of halo2's multiopen; GWC's per-point step):
// out_evals[j] = p_j(zeta), out_proof = commit((f - f(zeta)) / (X - zeta)) for f = sum_j v^j p_j -- the verifier folds the k
// commitments and evaluations with the same powers of v.  All k evaluations come out of one launch per level of the scan
// (block values only), f out of one pass over the k inputs; the opening of f is zkmi_bn254_kzg_open_dev's.
int32_t zkmi_bn254_kzg_open_many_dev(zkmi_ctx* ctx, const void* const* d_polys, uint32_t k, uint64_t n, const uint8_t zeta[32], const uint8_t v[32],
                                     const zkmi_bn_bases* srs, uint8_t* out_evals, uint8_t out_proof[64]) {
  ZK_ENTER(ctx);
  if (!d_polys || !zeta || !v || !srs || !out_evals || !out_proof || k == 0 || k > 4096 || n == 0 || n > MSM_MAX_TERMS || srs->n + 1 < n)
    return ZKMI_ERR_BAD_ARG;
  for (uint32_t j = 0; j < k; j++)
    if (!d_polys[j]) return ZKMI_ERR_BAD_ARG;
  if (!scalars_canonical(zeta, 1) || !scalars_canonical(v, 1)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "bn254 zeta / v >= r");
  uint32_t zw[8], vw[8];
  memcpy(zw, zeta, 32);
  memcpy(vw, v, 32);
  const uint64_t nb1 = (n + KZG_L - 1) / KZG_L, nb2 = (nb1 + KZG_L - 1) / KZG_L;
  // own allocation (the opening of f below re-sizes the context's staging and work buffers):
  // f[n] canonical | pointer table | pows | V1[k][nb1] | V2[k][nb2] | V3[k] | evals (k x 8 words)
  const size_t off_tab = 32 * n, off_pows = off_tab + 8 * (size_t)k, off_v1 = off_pows + sizeof(BnFr28) * KZG_POWS,
               off_v2 = off_v1 + sizeof(BnFr28) * k * nb1, off_v3 = off_v2 + sizeof(BnFr28) * k * nb2, off_ev = off_v3 + sizeof(BnFr28) * k;
  uint8_t* buf = nullptr;
  ZK_HIP(ctx, hipMalloc(&buf, off_ev + 32 * (size_t)k));
  uint32_t* d_f = reinterpret_cast<uint32_t*>(buf);
  const void** d_tab = reinterpret_cast<const void**>(buf + off_tab);
  BnFr28* pows = reinterpret_cast<BnFr28*>(buf + off_pows);
  BnFr28 *v1 = reinterpret_cast<BnFr28*>(buf + off_v1), *v2 = reinterpret_cast<BnFr28*>(buf + off_v2), *v3 = reinterpret_cast<BnFr28*>(buf + off_v3);
  uint32_t* d_ev = reinterpret_cast<uint32_t*>(buf + off_ev);
  const hipStream_t st = ctx->stream;
  hipError_t e = hipMemcpyAsync(d_tab, d_polys, 8 * (size_t)k, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_kzg_pow_table, dim3(1), dim3(64), 0, st, BnFr28::from_canonical(zw), pows);
    hipLaunchKernelGGL((k_kzg_scan<true, false, false>), dim3((unsigned)nb1, k), dim3(KZG_T), 0, st, nullptr, n, pows, 0, nullptr, 0u, v1, nullptr, nullptr,
                       reinterpret_cast<const void* const*>(d_tab), (uint64_t)0, nb1);
    const BnFr28* tops = v1;
    uint64_t top_stride = nb1;
    if (nb1 > 1) {
      hipLaunchKernelGGL((k_kzg_scan<false, false, false>), dim3((unsigned)nb2, k), dim3(KZG_T), 0, st, v1, nb1, pows, 10, nullptr, 0u, v2, nullptr, nullptr,
                         nullptr, nb1, nb2);
      tops = v2, top_stride = nb2;
      if (nb2 > 1) {
        hipLaunchKernelGGL((k_kzg_scan<false, false, false>), dim3(1, k), dim3(KZG_T), 0, st, v2, nb2, pows, 20, nullptr, 0u, v3, nullptr, nullptr, nullptr,
                           nb2, (uint64_t)1);
        tops = v3, top_stride = 1;
      }
    }
    hipLaunchKernelGGL(k_kzg_evals_out, dim3((k + 63) / 64), dim3(64), 0, st, tops, top_stride, k, d_ev);
    hipLaunchKernelGGL(k_kzg_lincomb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const void* const*>(d_tab), k, n,
                       BnFr28::from_canonical(vw), d_f);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_evals, d_ev, 32 * (size_t)k, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  int32_t rc = e == hipSuccess ? ZKMI_OK : ctx->hip_fail(e, "bn254 kzg open (many)");
  uint8_t f_eval[32];
  if (rc == ZKMI_OK) rc = kzg_open_core(ctx, d_f, n, zw, srs, nullptr, f_eval, out_proof);
  (void)hipFree(buf);
  return rc;
}

// Grand product of a permutation argument: d_out[0] = 1, d_out[i] = prod_{j < i} num_j / den_j (i < n), out_total = the
// product over all n (1 for a satisfied permutation).  Inputs and outputs: canonical words in HBM.  A zero denominator is an
// argument error (halo2's batch_invert would leave it zero and the argument would not verify).
int32_t zkmi_bn254_grand_product_dev(zkmi_ctx* ctx, const void* d_num, const void* d_den, uint64_t n, void* d_out, uint8_t out_total[32]) {
  ZK_ENTER(ctx);
  if (!d_num || !d_den || !d_out || !out_total || n == 0 || n > MSM_MAX_TERMS) return ZKMI_ERR_BAD_ARG;
  const uint64_t nb1 = (n + KZG_L - 1) / KZG_L, nb2 = (nb1 + KZG_L - 1) / KZG_L;
  // work: ratio[n] | V1[nb1] | E1[nb1] | V2[nb2] | E2[nb2] | total (8 words) | flag
  ZK_HIP(ctx, work_buffer(ctx, sizeof(BnFr28) * (n + 2 * nb1 + 2 * nb2 + 4)));
  BnFr28* ratio = static_cast<BnFr28*>(ctx->d_work);
  BnFr28 *v1 = ratio + n, *e1 = v1 + nb1, *v2 = e1 + nb1, *e2 = v2 + nb2;
  uint32_t* d_total = reinterpret_cast<uint32_t*>(e2 + nb2);
  uint32_t* d_flag = d_total + 8;
  const hipStream_t st = ctx->stream;
  PhaseTimer* t = ctx->timer();
  if (t) t->begin(PH_MISC, st);
  ZK_HIP(ctx, hipMemsetAsync(d_flag, 0, 4, st));
  hipLaunchKernelGGL(k_gp_ratio, dim3((unsigned)((n + 64 * GP_E - 1) / (64 * GP_E))), dim3(64), 0, st, d_num, d_den, n, ratio, d_flag);
  const BnFr28* carry1 = nullptr;
  if (nb1 > 1) {
    hipLaunchKernelGGL((k_gp_scan<false, false>), dim3((unsigned)nb1), dim3(KZG_T), 0, st, ratio, n, nullptr, v1, nullptr, nullptr);
    const BnFr28* carry2 = nullptr;
    if (nb2 > 1) {
      hipLaunchKernelGGL((k_gp_scan<false, false>), dim3((unsigned)nb2), dim3(KZG_T), 0, st, v1, nb1, nullptr, v2, nullptr, nullptr);
      hipLaunchKernelGGL((k_gp_scan<true, false>), dim3(1), dim3(KZG_T), 0, st, v2, nb2, nullptr, nullptr, e2, nullptr);
      carry2 = e2;
    }
    hipLaunchKernelGGL((k_gp_scan<true, false>), dim3((unsigned)nb2), dim3(KZG_T), 0, st, v1, nb1, carry2, nullptr, e1, nullptr);
    carry1 = e1;
  }
  hipLaunchKernelGGL((k_gp_scan<true, true>), dim3((unsigned)nb1), dim3(KZG_T), 0, st, ratio, n, carry1, nullptr, d_out, d_total);
  ZK_HIP(ctx, hipGetLastError());
  if (t) t->end(PH_MISC, st);
  uint32_t host[9];
  ZK_HIP(ctx, hipMemcpyAsync(host, d_total, 36, hipMemcpyDeviceToHost, st));
  ZK_HIP(ctx, hipStreamSynchronize(st));
  if (host[8]) return ctx->fail(ZKMI_ERR_BAD_ARG, "grand product: zero denominator");
  memcpy(out_total, host, 32);
  return ZKMI_OK;
}

}  // extern "C"

// zkmi — device-side Fq for the MSM kernels: 14 signed 28-bit limbs in 32-bit
// words ("unsaturated limbs"), Montgomery radix R = 2^392.
//
// Why this representation on gfx950 (measured with scripts/ubench.hip, see
// DESIGN.md): v_mad_i64_i32 / v_mad_u64_u32 issue at the same rate as a
// carry-propagating v_addc_co_u32, so with saturated 32-bit limbs every partial
// product costs two issue slots (multiply-add + carry) and hipcc additionally
// spends a v_mov per product assembling 64-bit addends (1329 instructions per
// 384-bit product).  With 28-bit limbs a 64-bit column accumulator absorbs all
// 28 partial products of a Montgomery multiplication with no carry handling at
// all: 196 + 196 multiply-adds and one carry sweep at the end (~500
// instructions), and field add/sub are 14 independent v_add/v_sub plus a
// 3-instruction-per-limb carry sweep.
//
// Invariants ("normalised"): limbs 0..12 in [0, 2^28); limb 13 signed and
// small; the represented integer v = sum l[i] 2^(28 i) satisfies |v| < 16 p.
// Values are NOT reduced to [0, p): a residue has several representations.
//   * mul/sqr accept any normalised operands (also one lazy add/sub of two
//     normalised values) and return v in (-p/2, 3p/2).
//   * is_zero() is exact for |v| <= 4p, which covers products, Fq2 products and
//     canonical inputs — the only places it is called (see curve.hpp).
//   * to_canonical() returns the unique representative in [0, p).
#pragma once
#include "field.hpp"

namespace zkmi {

struct Fq28Params {
  static constexpr int NL = 14;   // limbs
  static constexpr int N32 = 12;  // 32-bit words of the canonical form
  static constexpr uint32_t INV = 0xffcfffdu;  // -p^-1 mod 2^28
  static constexpr int32_t MOD[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2,
                                      0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x001a011};
  static constexpr int32_t ONE[14] = {0x347fcb8, 0xd800000, 0x002b119, 0x0cde6d2, 0xc7212e0, 0x83a2090, 0x037669f,
                                      0xda0f73e, 0x9b09b42, 0x1297bb0, 0x515d98f, 0x012ca7c, 0x659fcfa, 0x000577a};
  static constexpr int32_t R2[14] = {0x10370ed, 0x6d1c345, 0xe243d62, 0xec45c53, 0x3b1d65a, 0x093317d, 0xb4f36a0,
                                     0x5d74088, 0xc10ea72, 0x865d118, 0x7320a75, 0xfd5cd50, 0xcc8a759, 0x000c8d4};
};

struct Fr28Params {
  static constexpr int NL = 10;  // 280-bit radix, 25 spare bits
  static constexpr int N32 = 8;
  static constexpr uint32_t INV = 0xfffffffu;  // -r^-1 mod 2^28
  static constexpr int32_t MOD[10] = {0x0000001, 0xffffff0, 0xe5bfeff, 0xa402fff, 0x80553bd,
                                      0x0809a1d, 0x83339d8, 0x299d7d4, 0x3eda753, 0x0000007};
  static constexpr int32_t ONE[10] = {0xdcaaf6c, 0x355093f, 0x8209402, 0x41e37a6, 0x135587d,
                                      0x26172ba, 0x6854f56, 0x3973f39, 0xbc66e55, 0x0000006};
  static constexpr int32_t R2[10] = {0xc31bba9, 0x3b3440e, 0xe045fb0, 0x8929657, 0x57c6e1a,
                                     0x2d645cf, 0x012ecf5, 0xea6a1c5, 0xc7b9d12, 0x0000003};
  static constexpr uint32_t MOD32[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr int NUM_BITS = 255;
};

// BN254 scalar field (halo2curves::bn256::Fr, the field the reference's relations are written
// over: shielder/Cargo.lock:454-478); used by the Poseidon kernels only.
struct BnFr28Params {
  static constexpr int NL = 10;
  static constexpr int N32 = 8;
  static constexpr uint32_t INV = 0xfffffffu;
  static constexpr int32_t MOD[10] = {0x0000001, 0xe1f593f, 0x9709143, 0xe84879b, 0x85d2833,
                                      0xb681815, 0x9b85045, 0xe131a02, 0x0644e72, 0x0000003};
  static constexpr int32_t ONE[10] = {0xab5b8ba, 0xa771fc5, 0x1e556e4, 0x939ee8c, 0xdd60d0e,
                                      0x105a695, 0x89f6e5c, 0x8c59c9e, 0x7359fa8, 0x0000000};
  static constexpr int32_t R2[10] = {0xf4ec6b4, 0x9b2c977, 0x8aa70ff, 0x88742bb, 0xae1fcf6,
                                     0xff538b3, 0xbf33da0, 0xa1d99a8, 0xa45a5a6, 0x0000000};
  static constexpr uint32_t MOD32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr int NUM_BITS = 254;
};

template <class P>
struct Fp28 {
  static constexpr int NL = P::NL;
  static constexpr int32_t MASK = (1 << 28) - 1;
  int32_t l[NL];

  ZK_HD static Fp28 zero() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = 0;
    return r;
  }
  ZK_HD static Fp28 one() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = P::ONE[i];
    return r;
  }
  // signed carry sweep: limbs 0..NL-2 -> [0, 2^28), the top limb absorbs the rest
  ZK_HD void carry() {
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
      const int32_t c = l[i] >> 28;
      l[i] &= MASK;
      l[i + 1] += c;
    }
  }
  ZK_HD friend Fp28 operator+(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    r.carry();
    return r;
  }
  ZK_HD friend Fp28 operator-(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] - b.l[i];
    r.carry();
    return r;
  }
  // lazy forms: no carry sweep; result may only feed mul/sqr or one carry()
  ZK_HD Fp28 add_lazy(const Fp28& b) const {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = l[i] + b.l[i];
    return r;
  }
  ZK_HD Fp28 sub_lazy(const Fp28& b) const {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = l[i] - b.l[i];
    return r;
  }
  ZK_HD Fp28 neg() const {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = -l[i];
    r.carry();
    return r;
  }
  ZK_HD Fp28 dbl() const {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = l[i] * 2;
    r.carry();
    return r;
  }

  // Montgomery reduction of the 2*NL column accumulators, result from columns NL..
  ZK_HD static Fp28 reduce(int64_t* T) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int32_t m = (int32_t)(((uint32_t)T[k] * P::INV) & (uint32_t)MASK);
#pragma unroll
      for (int j = 0; j < NL; j++) T[k + j] += (int64_t)m * P::MOD[j];
      T[k + 1] += T[k] >> 28;  // exact: low 28 bits of T[k] are now zero
    }
    Fp28 r;
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; k++) {
      const int64_t v = T[NL + k] + c;
      r.l[k] = (int32_t)v & MASK;
      c = v >> 28;
    }
    r.l[NL - 1] = (int32_t)(T[2 * NL - 1] + c);
    return r;
  }

  ZK_HD friend Fp28 operator*(const Fp28& a, const Fp28& b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_CALL_MUL28)
    return mul_call(a, b);
#elif defined(ZK_FIPS28)
    return mul_fips(a, b);
#else
    return mul_inline(a, b);
#endif
  }
  ZK_HD Fp28 sqr() const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_CALL_MUL28)
    return sqr_call(*this);
#elif defined(ZK_FIPS28)
    return sqr_fips();
#else
    return sqr_inline();
#endif
  }
  // out-of-line device copies (one shared routine per TU; keeps G2 kernels
  // inside the instruction cache)
  __device__ __attribute__((noinline)) static Fp28 mul_call(Fp28 a, Fp28 b) { return mul_inline(a, b); }
  __device__ __attribute__((noinline)) static Fp28 sqr_call(Fp28 a) { return a.sqr_inline(); }
  ZK_HD static Fp28 mul_inline(const Fp28& a, const Fp28& b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_SETPRIO)
    // A/B build (make prio): raise the wave's issue priority for the multiply-add run of a product (round-3 experiment)
    struct Prio {
      __device__ Prio() { __builtin_amdgcn_s_setprio(2); }
      __device__ ~Prio() { __builtin_amdgcn_s_setprio(0); }
    } prio_guard;
#endif
    int64_t T[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
      for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
    return reduce(T);
  }
  ZK_HD Fp28 sqr_inline() const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_SETPRIO)
    struct Prio {
      __device__ Prio() { __builtin_amdgcn_s_setprio(2); }
      __device__ ~Prio() { __builtin_amdgcn_s_setprio(0); }
    } prio_guard;
#endif
    int64_t T[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      T[2 * i] += (int64_t)l[i] * l[i];
      const int32_t d = l[i] * 2;
#pragma unroll
      for (int j = i + 1; j < NL; j++) T[i + j] += (int64_t)d * l[j];
    }
    return reduce(T);
  }

  // ---- product-scanning forms ("finely integrated product scanning") -----------------------------
  // The same Montgomery product column by column: ONE 64-bit accumulator walks the 2 NL columns, the
  // reduction multiples m_k are produced as soon as their column is complete.  Same NL^2 + NL^2 multiply-adds
  // as mul_inline, but the live state is a, b, m (NL words) and one accumulator instead of 2 NL 64-bit
  // columns (56 registers for Fq): what lets the bucket-accumulation kernels keep a third wave per SIMD.
  // Column bound: <= 2 NL products of < 2^56 (+ lazy operands, see header) + carry < 2^62.
  // sum_{i+j=k} (a1[i] b1[j] + a2[i] b2[j]) with the second pair optional (lazy a b + c d under one reduction)
  // NP = number of product pairs summed under the one reduction (1, 2 or 4)
  template <int NP>
  ZK_HD static Fp28 fipsn(const Fp28* const* a, const Fp28* const* b) {
    int32_t m[NL];
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
#pragma unroll
        for (int q = 0; q < NP; q++) acc += (int64_t)a[q]->l[i] * b[q]->l[k - i];
      }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P::MOD[k - i];
      m[k] = (int32_t)(((uint32_t)acc * P::INV) & (uint32_t)MASK);
      acc += (int64_t)m[k] * P::MOD[0];
      acc >>= 28;  // exact: the low 28 bits are zero
    }
    Fp28 r;
#pragma unroll
    for (int k = NL; k < 2 * NL; k++) {
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++) {
#pragma unroll
        for (int q = 0; q < NP; q++) acc += (int64_t)a[q]->l[i] * b[q]->l[k - i];
      }
#pragma unroll
      for (int i = k - NL + 1; i < NL; i++) acc += (int64_t)m[i] * P::MOD[k - i];
      if (k < 2 * NL - 1) {
        r.l[k - NL] = (int32_t)acc & MASK;
        acc >>= 28;
      } else {
        r.l[NL - 1] = (int32_t)acc;
      }
    }
    return r;
  }
  template <bool TWO>
  ZK_HD static Fp28 fips(const Fp28& a1, const Fp28& b1, const Fp28& a2, const Fp28& b2) {
    const Fp28* a[2] = {&a1, &a2};
    const Fp28* b[2] = {&b1, &b2};
    return fipsn<TWO ? 2 : 1>(a, b);
  }
  ZK_HD static Fp28 mul_fips(const Fp28& a, const Fp28& b) { return fips<false>(a, b, a, b); }
  ZK_HD Fp28 sqr_fips() const {
    // squares: the symmetric terms once, doubled (d = 2 a_i fits 30 bits)
    int32_t m[NL];
    int64_t acc = 0;
    Fp28 r;
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) {
      const int lo = k < NL ? 0 : k - NL + 1, hi = k < NL ? k : NL - 1;
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        const int j = k - i;
        if (i < j) acc += (int64_t)(l[i] * 2) * l[j];
        else if (i == j) acc += (int64_t)l[i] * l[i];
      }
      if (k < NL) {
#pragma unroll
        for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P::MOD[k - i];
        m[k] = (int32_t)(((uint32_t)acc * P::INV) & (uint32_t)MASK);
        acc += (int64_t)m[k] * P::MOD[0];
        acc >>= 28;
      } else {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (int64_t)m[i] * P::MOD[k - i];
        if (k < 2 * NL - 1) {
          r.l[k - NL] = (int32_t)acc & MASK;
          acc >>= 28;
        } else {
          r.l[NL - 1] = (int32_t)acc;
        }
      }
    }
    return r;
  }

  // Fermat inverse x^(m-2) (0 -> 0); table construction only, never on the proving path
  __host__ __device__ Fp28 inv() const {
    int32_t e[NL];
    for (int i = 0; i < NL; i++) e[i] = P::MOD[i];
    e[0] -= 2;
    for (int i = 0; i < NL - 1; i++)
      if (e[i] < 0) {
        e[i] += (1 << 28);
        e[i + 1] -= 1;
      }
    Fp28 res = one();
    bool started = false;
    for (int i = NL - 1; i >= 0; i--)
      for (int b = 27; b >= 0; b--) {
        if (started) res = res.sqr();
        if ((e[i] >> b) & 1) {
          res = started ? res * (*this) : *this;
          started = true;
        }
      }
    return res;
  }

  // exact for |v| <= 4p (see header): v == k p for some |k| <= 4
  ZK_HD bool is_zero() const {
    // top limbs of the normalised representations of k*p, k = -4..4
    const int32_t t = l[NL - 1];
    bool cand = false;
#pragma unroll
    for (int k = -4; k <= 4; k++) cand |= (t == kp_limb(k, NL - 1));
    if (!cand) return false;
    bool hit = false;
#pragma unroll
    for (int k = -4; k <= 4; k++) {
      int32_t diff = 0;
#pragma unroll
      for (int i = 0; i < NL; i++) diff |= l[i] ^ kp_limb(k, i);
      hit |= (diff == 0);
    }
    return hit;
  }
  // limb i of the normalised representation of k*p (compile-time foldable)
  ZK_HD static constexpr int32_t kp_limb(int k, int i) {
    int64_t c = 0;
    int32_t out = 0;
    for (int j = 0; j <= i; j++) {
      const int64_t v = (int64_t)k * P::MOD[j] + c;
      if (j < NL - 1) {
        out = (int32_t)(v & MASK);
        c = v >> 28;
      } else {
        out = (int32_t)v;
      }
    }
    return out;
  }

  // canonical little-endian 32-bit words (plain integer < p) -> Montgomery limbs
  ZK_HD static Fp28 from_canonical(const uint32_t* w) {
    Fp28 a;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int bit = 28 * i;
      const int wi = bit >> 5, sh = bit & 31;
      uint64_t v = (wi < P::N32) ? w[wi] : 0u;
      if (wi + 1 < P::N32) v |= (uint64_t)w[wi + 1] << 32;
      a.l[i] = (int32_t)((v >> sh) & (uint32_t)MASK);
    }
    Fp28 r2;
#pragma unroll
    for (int i = 0; i < NL; i++) r2.l[i] = P::R2[i];
    return a * r2;
  }
  // Montgomery limbs -> canonical words in [0, p)
  ZK_HD void to_canonical(uint32_t* w) const {
    Fp28 o = zero();
    o.l[0] = 1;
    Fp28 c = (*this) * o;  // in [0, p]
    bool is_p = true;
#pragma unroll
    for (int i = 0; i < NL; i++) is_p &= (c.l[i] == P::MOD[i]);
#pragma unroll
    for (int i = 0; i < P::N32; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const uint64_t v = is_p ? 0u : (uint64_t)(uint32_t)c.l[i];
      const int bit = 28 * i;
      const int wi = bit >> 5, sh = bit & 31;
      const uint64_t s = v << sh;
      if (wi < P::N32) w[wi] |= (uint32_t)s;
      if (wi + 1 < P::N32) w[wi + 1] |= (uint32_t)(s >> 32);
    }
  }
};

using Fq28 = Fp28<Fq28Params>;
struct BnFq28Params {
  static constexpr int NL = 10;
  static constexpr int N32 = 8;
  static constexpr uint32_t INV = 0x4866389u;
  static constexpr int32_t MOD[10] = {0x87cfd47, 0x208c16d, 0x1ca8d3c, 0x6a91687, 0x85d9781,
                                      0xb681815, 0x9b85045, 0xe131a02, 0x0644e72, 0x0000003};
  static constexpr int32_t ONE[10] = {0xa0e0d96, 0x6a56a28, 0x56d8f97, 0x1e6c92b, 0xee608fc,
                                      0x10581c8, 0x89f6e5c, 0x8c59c9e, 0x7359fa8, 0x0000000};
  static constexpr int32_t R2[10] = {0x4693c46, 0xbb888f3, 0xe2ac0dd, 0x1c4bb9b, 0x3d9e1b9,
                                     0xc1a7aec, 0x2c83580, 0x3cb4fa2, 0x95e2ea9, 0x0000000};
  static constexpr uint32_t MOD32[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr int NUM_BITS = 254;
};
using BnFq28 = Fp28<BnFq28Params>;
using BnFr28 = Fp28<BnFr28Params>;
using Fr28 = Fp28<Fr28Params>;  // scalar field: NTT / witness map (values may grow to ~2^21 r between products)

// ---- Fq2 components with lazy reduction (overloads of field.hpp's generic forms) ----
// a0 b0 + s a1 b1 summed in the 64-bit columns: 28 products of < 2^56 plus the
// reduction's 14 stay below 2^62; one reduction for the pair.
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZK_CALL_MUL28)
#define ZK_FQ2_28 __device__ __attribute__((noinline))
#else
#define ZK_FQ2_28 ZK_HD
#endif
ZK_FQ2_28 Fq28 fq2_mul_c0(const Fq28& a0, const Fq28& a1, const Fq28& b0, const Fq28& b1) {
  constexpr int NL = Fq28::NL;
  int64_t T[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t n1 = -a1.l[i];
#pragma unroll
    for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a0.l[i] * b0.l[j] + (int64_t)n1 * b1.l[j];
  }
  return Fq28::reduce(T);
}
ZK_FQ2_28 Fq28 fq2_mul_c1(const Fq28& a0, const Fq28& a1, const Fq28& b0, const Fq28& b1) {
  constexpr int NL = Fq28::NL;
  int64_t T[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++)
#pragma unroll
    for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a0.l[i] * b1.l[j] + (int64_t)a1.l[i] * b0.l[j];
  return Fq28::reduce(T);
}
ZK_FQ2_28 Fq28 fq2_sqr_c0(const Fq28& a0, const Fq28& a1) {
  return Fq28::mul_inline(a0.add_lazy(a1), a0.sub_lazy(a1));  // operands < 2^29 per limb
}
ZK_FQ2_28 Fq28 fq2_sqr_c1(const Fq28& a0, const Fq28& a1) {
  return Fq28::mul_inline(a0.add_lazy(a0), a1);
}

using Fq2_28 = Fq2T<Fq28>;

// ---- XYZZ building blocks (overloads of curve.hpp's generic forms) ------------
// Fq28: differences that only feed products skip the carry sweep (limbs < 2^29,
// 14 products of < 2^59 stay below 2^63 together with the reduction's share).
// (templates on the parameter set: the BLS12-381 and the BN254 base fields share them)
template <class P>
ZK_HD Fp28<P> f_sub_lazy(const Fp28<P>& a, const Fp28<P>& b) { return a.sub_lazy(b); }
// (+-a) - b without a carry sweep: m = 0 keeps a, m = 0xffffffff negates it (two's complement per limb).  The bucket
// accumulation folds the digit's sign into the first difference of the mixed addition instead of negating the
// point's y coordinate beforehand (a 14-limb negation plus a 39-instruction carry sweep per insertion).
template <class P>
ZK_HD Fp28<P> f_signed_sub_lazy(const Fp28<P>& a, uint32_t m, const Fp28<P>& b) {
  Fp28<P> r;
#pragma unroll
  for (int i = 0; i < P::NL; i++) r.l[i] = (int32_t)(((uint32_t)a.l[i] ^ m) - m) - b.l[i];
  return r;
}
template <class P>
ZK_HD Fp28<P> f_x3(const Fp28<P>& rr, const Fp28<P>& ppp, const Fp28<P>& q) {
  Fp28<P> r;
#pragma unroll
  for (int i = 0; i < P::NL; i++) r.l[i] = rr.l[i] - ppp.l[i] - 2 * q.l[i];
  r.carry();
  return r;
}
// a b - c d under one reduction (a, b may be lazy differences; c, d normalised)
template <class P>
ZK_HD Fp28<P> f_mul_sub_mul(const Fp28<P>& a, const Fp28<P>& b, const Fp28<P>& c, const Fp28<P>& d) {
  constexpr int NL = P::NL;
#if defined(ZK_FIPS28)
  {
    Fp28<P> nc;
#pragma unroll
    for (int i = 0; i < NL; i++) nc.l[i] = -c.l[i];
    return Fp28<P>::template fips<true>(a, b, nc, d);
  }
#endif
  int64_t T[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t nc = -c.l[i];
#pragma unroll
    for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * b.l[j] + (int64_t)nc * d.l[j];
  }
  return Fp28<P>::reduce(T);
}
// Fq2 over the limbs: operands stay normalised (a component already sums two
// products), but a b - c d needs only one reduction per component (4 x 14 products
// of < 2^56 plus the reduction stay below 2^62).
ZK_HD Fq2_28 f_x3(const Fq2_28& rr, const Fq2_28& ppp, const Fq2_28& q) {
  return {f_x3(rr.c0, ppp.c0, q.c0), f_x3(rr.c1, ppp.c1, q.c1)};
}
ZK_HD Fq2_28 f_mul_sub_mul(const Fq2_28& a, const Fq2_28& b, const Fq2_28& c, const Fq2_28& d) {
  constexpr int NL = Fq28::NL;
  Fq2_28 out;
  {
    int64_t T[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int32_t na1 = -a.c1.l[i], nc0 = -c.c0.l[i];
#pragma unroll
      for (int j = 0; j < NL; j++)
        T[i + j] += (int64_t)a.c0.l[i] * b.c0.l[j] + (int64_t)na1 * b.c1.l[j] + (int64_t)nc0 * d.c0.l[j] +
                    (int64_t)c.c1.l[i] * d.c1.l[j];
    }
    out.c0 = Fq28::reduce(T);
  }
  {
    int64_t T[2 * NL];
#pragma unroll
    for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int32_t nc0 = -c.c0.l[i], nc1 = -c.c1.l[i];
#pragma unroll
      for (int j = 0; j < NL; j++)
        T[i + j] += (int64_t)a.c0.l[i] * b.c1.l[j] + (int64_t)a.c1.l[i] * b.c0.l[j] + (int64_t)nc0 * d.c1.l[j] +
                    (int64_t)nc1 * d.c0.l[j];
    }
    out.c1 = Fq28::reduce(T);
  }
  return out;
}

// ---- Fq2 split across a lane pair (device only) ----------------------------------
// A full Fq2 mixed addition needs ~330 registers per lane (1 wave per SIMD).  Here the
// two components of every Fq2 value live in two adjacent lanes (even lane: c0, odd
// lane: c1), the partner's limbs are fetched with DPP quad_perm(1,0,3,2) moves when a
// product needs them, and each lane computes one component with ONE Montgomery
// reduction.  State per lane is that of a G1 addition (2 waves per SIMD), the pair
// does the same number of partial products as the unsplit form.
#if defined(__HIPCC__)
struct Fq2P {
  Fq28 v;  // this lane's component
  __device__ __forceinline__ static bool odd() { return (threadIdx.x & 1u) != 0; }
  __device__ __forceinline__ static Fq28 partner(const Fq28& a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], 0xB1, 0xF, 0xF, true);
    return r;
  }
  __device__ __forceinline__ static Fq28 sel(bool c, const Fq28& a, const Fq28& b) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
  }
  __device__ __forceinline__ static Fq28 negl(const Fq28& a) {  // lazy negation (limb-wise)
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) r.l[i] = -a.l[i];
    return r;
  }
  __device__ __forceinline__ static Fq2P zero() { return {Fq28::zero()}; }
  __device__ __forceinline__ static Fq2P one() { return {odd() ? Fq28::zero() : Fq28::one()}; }
  __device__ __forceinline__ bool is_zero() const {
    const int mine = v.is_zero() ? 1 : 0;
    const int other = __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xF, 0xF, true);
    return (mine & other) != 0;
  }
  __device__ __forceinline__ friend Fq2P operator+(const Fq2P& a, const Fq2P& b) { return {a.v + b.v}; }
  __device__ __forceinline__ friend Fq2P operator-(const Fq2P& a, const Fq2P& b) { return {a.v - b.v}; }
  __device__ __forceinline__ Fq2P neg() const { return {v.neg()}; }
  __device__ __forceinline__ Fq2P dbl() const { return {v.dbl()}; }
  // component = a.v * X + partner(a) * Y with
  //   even lane (c0 = a0 b0 - a1 b1): X = b.v,        Y = -partner(b)
  //   odd  lane (c1 = a1 b0 + a0 b1): X = partner(b), Y = b.v
  __device__ __forceinline__ static void mul_cols(int64_t* T, const Fq2P& a, const Fq2P& b, bool negate) {
    constexpr int NL = Fq28::NL;
    const bool o = odd();
    // two sequential product phases keep only one partner copy + one selected operand live
    {
      const Fq28 pb = partner(b.v);
      Fq28 X = sel(o, pb, b.v);
      if (negate) X = negl(X);
#pragma unroll
      for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.v.l[i] * X.l[j];
    }
    {
      const Fq28 pa = partner(a.v);
      Fq28 Y = sel(o, b.v, negl(partner(b.v)));
      if (negate) Y = negl(Y);
#pragma unroll
      for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) T[i + j] += (int64_t)pa.l[i] * Y.l[j];
    }
  }
  // product-scanning form of the same component: operands of the two (four) products first, then one
  // accumulator walks the columns (field28.hpp fipsn) -- no 2 NL 64-bit columns live
  __device__ __forceinline__ static void operands(const Fq2P& a, const Fq2P& b, bool negate, Fq28& X, Fq28& pa, Fq28& Y) {
    const bool o = odd();
    const Fq28 pb = partner(b.v);
    X = sel(o, pb, b.v);
    Y = sel(o, b.v, negl(pb));
    if (negate) {
      X = negl(X);
      Y = negl(Y);
    }
    pa = partner(a.v);
  }
  __device__ __forceinline__ friend Fq2P operator*(const Fq2P& a, const Fq2P& b) {
#if defined(ZK_FIPS28)
    {
      Fq28 X, pa, Y;
      operands(a, b, false, X, pa, Y);
      return {Fq28::fips<true>(a.v, X, pa, Y)};
    }
#endif
    int64_t T[2 * Fq28::NL];
#pragma unroll
    for (int i = 0; i < 2 * Fq28::NL; i++) T[i] = 0;
    mul_cols(T, a, b, false);
    return {Fq28::reduce(T)};
  }
  // even: (a0 + a1)(a0 - a1); odd: (2 a0) a1   — one product per lane
  __device__ __forceinline__ Fq2P sqr() const {
    const bool o = odd();
    const Fq28 p = partner(v);
    const Fq28 X = sel(o, p.add_lazy(p), v.add_lazy(p));
    const Fq28 Y = sel(o, v, v.sub_lazy(p));
    return {Fq28::mul_inline(X, Y)};
  }
};
// a b - c d, one reduction per lane (4 x 14 products of < 2^56 per column)
__device__ __forceinline__ Fq2P f_mul_sub_mul(const Fq2P& a, const Fq2P& b, const Fq2P& c, const Fq2P& d) {
#if defined(ZK_FIPS28)
  {
    Fq28 X1, pa, Y1, X2, pc, Y2;
    Fq2P::operands(a, b, false, X1, pa, Y1);
    Fq2P::operands(c, d, true, X2, pc, Y2);
    const Fq28* A[4] = {&a.v, &pa, &c.v, &pc};
    const Fq28* B[4] = {&X1, &Y1, &X2, &Y2};
    return {Fq28::fipsn<4>(A, B)};
  }
#endif
  int64_t T[2 * Fq28::NL];
#pragma unroll
  for (int i = 0; i < 2 * Fq28::NL; i++) T[i] = 0;
  Fq2P::mul_cols(T, a, b, false);
  Fq2P::mul_cols(T, c, d, true);
  return {Fq28::reduce(T)};
}
// (+-a) - b per component; carried (unlike the Fq form above): an Fq2 product column sums two partial products, so its
// operands have no spare bit for a lazy difference.  Still one carry sweep instead of the two of "negate, then subtract".
__device__ __forceinline__ Fq2P f_signed_sub_lazy(const Fq2P& a, uint32_t m, const Fq2P& b) {
  Fq28 r;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) r.l[i] = (int32_t)(((uint32_t)a.v.l[i] ^ m) - m) - b.v.l[i];
  r.carry();
  return {r};
}
__device__ __forceinline__ Fq2P f_x3(const Fq2P& rr, const Fq2P& ppp, const Fq2P& q) {
  return {f_x3(rr.v, ppp.v, q.v)};
}
#endif  // __HIPCC__

// conversions between the host/old representation (12x32, R = 2^384) and Fq28
ZK_HD Fq28 fq28_from_fq(const Fq& a) {
  Fq c = a.from_mont();
  return Fq28::from_canonical(c.l);
}
ZK_HD Fq fq_from_fq28(const Fq28& a) {
  Fq c;
  a.to_canonical(c.l);
  return c.to_mont();
}
ZK_HD BnFq28 fq28_from_fq(const BnFq& a) {
  BnFq c = a.from_mont();
  return BnFq28::from_canonical(c.l);
}
ZK_HD BnFq fq_from_fq28(const BnFq28& a) {
  BnFq c;
  a.to_canonical(c.l);
  return c.to_mont();
}
ZK_HD BnFr28 fr28_from_fr(const BnFr& a) {
  BnFr c = a.from_mont();
  return BnFr28::from_canonical(c.l);
}
ZK_HD Fq2_28 fq28_from_fq(const Fq2& a) { return {fq28_from_fq(a.c0), fq28_from_fq(a.c1)}; }
ZK_HD Fq2 fq_from_fq28(const Fq2_28& a) { return {fq_from_fq28(a.c0), fq_from_fq28(a.c1)}; }

}  // namespace zkmi

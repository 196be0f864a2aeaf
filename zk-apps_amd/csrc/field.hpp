// zkmi — Montgomery prime-field arithmetic for BLS12-381 (Fr: 8x32-bit limbs,
// Fq: 12x32-bit limbs), shared by gfx950 device code and the host-side O(1)
// steps (proof assembly, window combine, verifier).
//
// Reference locus: none in /root/reference; the
// arithmetic restates the published BLS12-381 field definition.  Elements are
// little-endian limb arrays in Montgomery form (x*2^(32N) mod m), always fully
// reduced to [0, m), so equal field elements have equal bytes.
//
// Why 32-bit limbs: CDNA4's widest integer multiply is v_mad_u64_u32
// (32x32+64 -> 64); 64-bit limbs would only be split by the compiler.  No MFMA:
// these are carry-chained modular products, not dense contractions.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define ZK_HD __host__ __device__ __forceinline__

namespace zkmi {

struct FrParams {
  static constexpr int N = 8;
  static constexpr uint32_t INV = 0xffffffffu;  // -m^-1 mod 2^32
  static constexpr uint64_t INV64 = 0xfffffffeffffffffull;
  static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                      0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                      0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

struct FqParams {
  static constexpr int N = 12;
  static constexpr uint32_t INV = 0xfffcfffdu;
  static constexpr uint64_t INV64 = 0x89f3fffcfffcfffdull;
  static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                       0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                       0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                       0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                       0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                      0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                      0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

// BN254 (alt_bn128): the curve of the reference's own proving stack (halo2curves::bn256,
// shielder/Cargo.lock:454-478).  Used by the KZG-commit-shaped MSM / NTT driver.
struct BnFqParams {
  static constexpr int N = 8;
  static constexpr uint32_t INV = 0xe4866389u;
  static constexpr uint64_t INV64 = 0x87d20782e4866389ull;
  static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct BnFrParams {
  static constexpr int N = 8;
  static constexpr uint32_t INV = 0xefffffffu;
  static constexpr uint64_t INV64 = 0xc2e1f593efffffffull;
  static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class P>
struct Fp {
  static constexpr int N = P::N;
  uint32_t l[N];

  ZK_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  ZK_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
    return r;
  }
  ZK_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::R2[i];
    return r;
  }
  ZK_HD bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= l[i];
    return acc == 0;
  }
  ZK_HD bool operator==(const Fp& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
    return acc == 0;
  }
  ZK_HD bool operator!=(const Fp& o) const { return !(*this == o); }

  // r = a - m if a >= m (a < 2m assumed)
  ZK_HD static void cond_sub_mod(uint32_t* a, uint32_t top) {
    uint32_t t[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t d = (uint64_t)a[i] - P::MOD[i] - borrow;
      t[i] = (uint32_t)d;
      borrow = (d >> 63) & 1;
    }
    // subtract succeeded if no final borrow, or the (N+1)-th word absorbs it
    bool ge = (top != 0) || (borrow == 0);
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = ge ? t[i] : a[i];
  }

  ZK_HD friend Fp operator+(const Fp& a, const Fp& b) {
    Fp r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t s = (uint64_t)a.l[i] + b.l[i] + carry;
      r.l[i] = (uint32_t)s;
      carry = s >> 32;
    }
    cond_sub_mod(r.l, (uint32_t)carry);
    return r;
  }
  ZK_HD friend Fp operator-(const Fp& a, const Fp& b) {
    Fp r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
      r.l[i] = (uint32_t)d;
      borrow = (d >> 63) & 1;
    }
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t s = (uint64_t)r.l[i] + (P::MOD[i] & mask) + carry;
      r.l[i] = (uint32_t)s;
      carry = s >> 32;
    }
    return r;
  }
  ZK_HD Fp neg() const {
    if (is_zero()) return *this;
    Fp r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t d = (uint64_t)P::MOD[i] - l[i] - borrow;
      r.l[i] = (uint32_t)d;
      borrow = (d >> 63) & 1;
    }
    return r;
  }
  ZK_HD Fp dbl() const { return *this + *this; }

  // CIOS Montgomery product: a*b*2^(-32N) mod m.  The moduli have spare top
  // bits (Fr: 1, Fq: 3) so the running value fits N+1 words.
  ZK_HD friend Fp operator*(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: same CIOS on 64-bit limbs (identical little-endian byte layout)
    constexpr int M = N / 2;
    uint64_t A[M], B[M], Q[M], u[M + 2];
    for (int i = 0; i < M; i++) {
      A[i] = a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
      B[i] = b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
      Q[i] = P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    for (int i = 0; i < M + 2; i++) u[i] = 0;
    for (int i = 0; i < M; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < M; j++) {
        unsigned __int128 s = (unsigned __int128)A[j] * B[i] + u[j] + (uint64_t)c;
        u[j] = (uint64_t)s;
        c = s >> 64;
      }
      unsigned __int128 s = (unsigned __int128)u[M] + (uint64_t)c;
      u[M] = (uint64_t)s;
      u[M + 1] = (uint64_t)(s >> 64);
      const uint64_t m = u[0] * P::INV64;
      s = (unsigned __int128)m * Q[0] + u[0];
      c = s >> 64;
      for (int j = 1; j < M; j++) {
        s = (unsigned __int128)m * Q[j] + u[j] + (uint64_t)c;
        u[j - 1] = (uint64_t)s;
        c = s >> 64;
      }
      s = (unsigned __int128)u[M] + (uint64_t)c;
      u[M - 1] = (uint64_t)s;
      u[M] = u[M + 1] + (uint64_t)(s >> 64);
    }
    Fp r;
    for (int i = 0; i < M; i++) {
      r.l[2 * i] = (uint32_t)u[i];
      r.l[2 * i + 1] = (uint32_t)(u[i] >> 32);
    }
    cond_sub_mod(r.l, (uint32_t)u[M]);
    return r;
#elif defined(ZK_CALL_MUL)
    return mul_call(a, b);
#else
    return mul_inline(a, b);
#endif
  }
  // Out-of-line device copy: one shared routine per TU instead of one inlined
  // body per call site (keeps G2 kernels inside the instruction cache).
  __device__ __attribute__((noinline)) static Fp mul_call(Fp a, Fp b) { return mul_inline(a, b); }
  __device__ __forceinline__ static Fp mul_inline(const Fp& a, const Fp& b) {
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      const uint32_t bi = b.l[i];
#pragma unroll
      for (int j = 0; j < N; j++) {
        uint64_t s = (uint64_t)a.l[j] * bi + t[j] + c;
        t[j] = (uint32_t)s;
        c = s >> 32;
      }
      uint64_t s = (uint64_t)t[N] + c;
      t[N] = (uint32_t)s;
      t[N + 1] = (uint32_t)(s >> 32);
      const uint32_t m = t[0] * P::INV;
      s = (uint64_t)m * P::MOD[0] + t[0];
      c = s >> 32;
#pragma unroll
      for (int j = 1; j < N; j++) {
        s = (uint64_t)m * P::MOD[j] + t[j] + c;
        t[j - 1] = (uint32_t)s;
        c = s >> 32;
      }
      s = (uint64_t)t[N] + c;
      t[N - 1] = (uint32_t)s;
      t[N] = t[N + 1] + (uint32_t)(s >> 32);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    cond_sub_mod(r.l, t[N]);
    return r;
  }
  ZK_HD Fp sqr() const { return (*this) * (*this); }

  ZK_HD Fp to_mont() const { return (*this) * r2(); }
  ZK_HD Fp from_mont() const {
    Fp o = zero();
    o.l[0] = 1;
    return (*this) * o;
  }

  // x^e for a little-endian 32-bit-limb exponent (not constant time; public data)
  __host__ __device__ Fp pow(const uint32_t* e, int nlimbs) const {
    Fp res = one();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        if (started) res = res.sqr();
        if ((e[i] >> b) & 1) {
          res = started ? res * (*this) : *this;
          started = true;
        }
      }
    }
    return res;
  }
  // Fermat inverse; returns 0 for 0
  __host__ __device__ Fp inv() const {
    uint32_t e[N];
    uint64_t borrow = 2;
    for (int i = 0; i < N; i++) {
      uint64_t d = (uint64_t)P::MOD[i] - borrow;
      e[i] = (uint32_t)d;
      borrow = (d >> 63) & 1;
    }
    return pow(e, N);
  }
  // canonical integer comparison a > (m-1)/2, used by the compressed encoding
  __host__ bool lex_larger() const {
    Fp c = from_mont();
    // compare 2c > m-1  <=> 2c >= m  (m odd)
    uint64_t carry = 0;
    uint32_t d[N + 1];
    for (int i = 0; i < N; i++) {
      uint64_t s = ((uint64_t)c.l[i] << 1) | carry;
      d[i] = (uint32_t)s;
      carry = s >> 32;
    }
    d[N] = (uint32_t)carry;
    if (d[N]) return true;
    for (int i = N - 1; i >= 0; i--) {
      if (d[i] != P::MOD[i]) return d[i] > P::MOD[i];
    }
    return true;
  }
};

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;
using BnFq = Fp<BnFqParams>;
using BnFr = Fp<BnFrParams>;

// Fq2 product/square components; the device limb representation overloads these
// (field28.hpp) to sum partial products in the column accumulators and pay one
// Montgomery reduction per component instead of one per base-field product.
template <class B>
ZK_HD B fq2_mul_c0(const B& a0, const B& a1, const B& b0, const B& b1) { return a0 * b0 - a1 * b1; }
template <class B>
ZK_HD B fq2_mul_c1(const B& a0, const B& a1, const B& b0, const B& b1) {
  return (a0 + a1) * (b0 + b1) - a0 * b0 - a1 * b1;
}
template <class B>
ZK_HD B fq2_sqr_c0(const B& a0, const B& a1) { return (a0 + a1) * (a0 - a1); }
template <class B>
ZK_HD B fq2_sqr_c1(const B& a0, const B& a1) { return (a0 * a1).dbl(); }

// Fq2 = Fq[u]/(u^2+1), generic over the base-field representation
template <class B>
struct Fq2T {
  B c0, c1;
  ZK_HD static Fq2T zero() { return {B::zero(), B::zero()}; }
  ZK_HD static Fq2T one() { return {B::one(), B::zero()}; }
  ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_HD bool operator==(const Fq2T& o) const { return c0 == o.c0 && c1 == o.c1; }
  ZK_HD bool operator!=(const Fq2T& o) const { return !(*this == o); }
  ZK_HD friend Fq2T operator+(const Fq2T& a, const Fq2T& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  ZK_HD friend Fq2T operator-(const Fq2T& a, const Fq2T& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  ZK_HD Fq2T neg() const { return {c0.neg(), c1.neg()}; }
  ZK_HD Fq2T dbl() const { return {c0.dbl(), c1.dbl()}; }
  ZK_HD friend Fq2T operator*(const Fq2T& a, const Fq2T& b) {
    return {fq2_mul_c0(a.c0, a.c1, b.c0, b.c1), fq2_mul_c1(a.c0, a.c1, b.c0, b.c1)};
  }
  ZK_HD Fq2T sqr() const { return {fq2_sqr_c0(c0, c1), fq2_sqr_c1(c0, c1)}; }
  ZK_HD Fq2T mul_fq(const B& k) const { return {c0 * k, c1 * k}; }
  ZK_HD Fq2T conj() const { return {c0, c1.neg()}; }
  // multiply by xi = 1 + u
  ZK_HD Fq2T mul_xi() const { return {c0 - c1, c0 + c1}; }
  __host__ __device__ Fq2T inv() const {
    B d = (c0.sqr() + c1.sqr()).inv();
    return {c0 * d, (c1 * d).neg()};
  }
  ZK_HD Fq2T to_mont() const { return {c0.to_mont(), c1.to_mont()}; }
  ZK_HD Fq2T from_mont() const { return {c0.from_mont(), c1.from_mont()}; }
  __host__ bool lex_larger() const {
    if (!c1.is_zero()) return c1.lex_larger();
    return c0.lex_larger();
  }
};
using Fq2 = Fq2T<Fq>;

}  // namespace zkmi

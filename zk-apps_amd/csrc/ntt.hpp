// zkmi — host-side handle for one radix-2 evaluation domain resident in HBM.
#pragma once
#include "field28.hpp"

namespace zkmi {

// 10-limb elements (BLS12-381 Fr, BN254 Fr): 40 bytes = 5 x 8-byte accesses
template <class F>
__device__ __forceinline__ F ld28(const F* p) {
  static_assert(F::NL == 10, "10-limb scalar fields");
  F r;
  const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint2 v = q[i];
    r.l[2 * i] = (int32_t)v.x;
    r.l[2 * i + 1] = (int32_t)v.y;
  }
  return r;
}
template <class F>
__device__ __forceinline__ void st28(F* p, const F& v) {
  uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; i++) q[i] = make_uint2((uint32_t)v.l[2 * i], (uint32_t)v.l[2 * i + 1]);
}

// Per-field constants of the radix-2 domains (host side)
template <class F> struct NttField;
template <> struct NttField<Fr28> {   // BLS12-381 Fr: two-adicity 32, generator 7 (ark-bls12-381)
  using Host = Fr;
  static constexpr int TWO_ADICITY = 32;
  static Host root_max();            // 7^((r-1)/2^32)
};
template <> struct NttField<BnFr28> { // BN254 Fr: two-adicity 28, generator 7 (halo2curves bn256::Fr)
  using Host = BnFr;
  static constexpr int TWO_ADICITY = 28;
  static Host root_max();            // 7^((r-1)/2^28) = halo2curves' ROOT_OF_UNITY
};

template <class F>
struct NttDomainT {
  int log_n = 0;
  F* tw_fwd = nullptr;           // w^k, k < N/2
  F* tw_inv = nullptr;           // w^-k
  F* coset_fwd = nullptr;        // g^i            (natural order, g = 7)
  F* coset_inv_n = nullptr;      // N^-1 g^-i      (natural order)
  F* rev_coset_n = nullptr;      // N^-1 g^rev(p)  (position-indexed, bit-reversed coefficients)
  F* rev_coset_inv_n = nullptr;  // N^-1 g^-rev(p)
  F* n_inv = nullptr;            // N^-1
  F* scratch = nullptr;          // N elements
  F n_inv_host;
  ~NttDomainT();
  hipError_t init(int log_n, hipStream_t stream);
  // natural order in and out (public entry point)
  hipError_t transform(F* d_data, bool inverse, bool coset, hipStream_t stream);
  // prover building blocks, no bit-reversal copies:
  //   inverse_to_rev : evaluations (natural) -> coefficients in bit-reversed order, each
  //                    multiplied by post_table[position]; optionally written as canonical words
  //   forward_from_rev: coefficients in bit-reversed order -> evaluations (natural)
  // batch > 1: `batch` vectors of 2^log_n elements back to back in d (and in canon_out), one launch per pass
  hipError_t inverse_to_rev(F* d, const F* post_table, uint32_t* canon_out, hipStream_t st, uint32_t batch = 1);
  hipError_t forward_from_rev(F* d, hipStream_t st, uint32_t batch = 1);
};

using NttDomain = NttDomainT<Fr28>;      // the prover's domain
using NttDomainBn = NttDomainT<BnFr28>;  // KZG-commit-shaped driver over BN254

Fr fr_root_of_unity(int log_n);
template <class F> hipError_t ntt_from_canonical(const uint32_t* d_in, F* d_out, uint32_t n, hipStream_t s);
template <class F> hipError_t ntt_to_canonical(const F* d_in, uint32_t* d_out, uint32_t n, hipStream_t s);
template <class F> hipError_t ntt_mul_table(F* d, const F* table, uint32_t n, hipStream_t s);
hipError_t ntt_enable_big_lds();

}  // namespace zkmi

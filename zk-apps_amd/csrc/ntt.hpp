// zkmi — host-side handle for one radix-2 evaluation domain resident in HBM.
#pragma once
#include "field28.hpp"

namespace zkmi {

__device__ __forceinline__ Fr28 ld28(const Fr28* p) {
  Fr28 r;
  const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint2 v = q[i];
    r.l[2 * i] = (int32_t)v.x;
    r.l[2 * i + 1] = (int32_t)v.y;
  }
  return r;
}
__device__ __forceinline__ void st28(Fr28* p, const Fr28& v) {
  uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
  for (int i = 0; i < 5; i++) q[i] = make_uint2((uint32_t)v.l[2 * i], (uint32_t)v.l[2 * i + 1]);
}

struct NttDomain {
  int log_n = 0;
  Fr28* tw_fwd = nullptr;           // w^k, k < N/2
  Fr28* tw_inv = nullptr;           // w^-k
  Fr28* coset_fwd = nullptr;        // g^i            (natural order, g = 7)
  Fr28* coset_inv_n = nullptr;      // N^-1 g^-i      (natural order)
  Fr28* rev_coset_n = nullptr;      // N^-1 g^rev(p)  (position-indexed, bit-reversed coefficients)
  Fr28* rev_coset_inv_n = nullptr;  // N^-1 g^-rev(p)
  Fr28* n_inv = nullptr;            // N^-1
  Fr28* scratch = nullptr;          // N elements
  Fr28 n_inv_host;
  ~NttDomain();
  hipError_t init(int log_n, hipStream_t stream);
  // natural order in and out (public entry point)
  hipError_t transform(Fr28* d_data, bool inverse, bool coset, hipStream_t stream);
  // prover building blocks, no bit-reversal copies:
  //   inverse_to_rev : evaluations (natural) -> coefficients in bit-reversed order, each
  //                    multiplied by post_table[position]; optionally written as canonical words
  //   forward_from_rev: coefficients in bit-reversed order -> evaluations (natural)
  hipError_t inverse_to_rev(Fr28* d, const Fr28* post_table, uint32_t* canon_out, hipStream_t st);
  hipError_t forward_from_rev(Fr28* d, hipStream_t st);
};

Fr fr_root_of_unity(int log_n);
hipError_t ntt_from_canonical(const uint32_t* d_in, Fr28* d_out, uint32_t n, hipStream_t s);
hipError_t ntt_to_canonical(const Fr28* d_in, uint32_t* d_out, uint32_t n, hipStream_t s);
hipError_t ntt_mul_table(Fr28* d, const Fr28* table, uint32_t n, hipStream_t s);
hipError_t ntt_enable_big_lds();

}  // namespace zkmi

// zkmi — host-side handle for one radix-2 evaluation domain resident in HBM.
#pragma once
#include "field.hpp"

namespace zkmi {

struct NttDomain {
  int log_n = 0;
  Fr* tw_fwd = nullptr;     // w^k, k < N/2
  Fr* tw_inv = nullptr;     // w^-k
  Fr* coset_fwd = nullptr;  // g^i, i < N (g = 7)
  Fr* coset_inv = nullptr;  // g^-i
  Fr* n_inv = nullptr;      // N^-1
  Fr* scratch = nullptr;    // N elements
  ~NttDomain();
  hipError_t init(int log_n, hipStream_t stream);
  hipError_t transform(Fr* d_data, bool inverse, bool coset, hipStream_t stream);
};

Fr fr_root_of_unity(int log_n);
hipError_t ntt_to_mont(Fr* d, uint32_t n, hipStream_t s);
hipError_t ntt_from_mont(Fr* d, uint32_t n, hipStream_t s);
hipError_t ntt_enable_big_lds();

}  // namespace zkmi

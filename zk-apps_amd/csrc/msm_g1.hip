// zkmi — G1 (F = Fq) instantiation of the Pippenger MSM kernels (msm_impl.hpp).
#include "msm_impl.hpp"
namespace zkmi {
template struct MsmEngine<Fq>;
template hipError_t bases_to_mont<Fq>(Affine<Fq>*, uint64_t, hipStream_t);
template XYZZ<Fq> msm_combine_windows<Fq>(const XYZZ<Fq>*, int, int);
}  // namespace zkmi

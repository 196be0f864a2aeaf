// zkmi — G2 (F = Fq2) instantiation of the Pippenger MSM kernels (msm_impl.hpp).
#define ZK_CALL_MUL 1
#include "msm_impl.hpp"
namespace zkmi {
template struct MsmEngine<Fq2>;
template hipError_t bases_to_mont<Fq2>(Affine<Fq2>*, uint64_t, hipStream_t);
template XYZZ<Fq2> msm_combine_windows<Fq2>(const XYZZ<Fq2>*, int, int);
}  // namespace zkmi

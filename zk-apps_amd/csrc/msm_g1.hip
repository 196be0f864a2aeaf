// zkmi — G1 (F = Fq28 limbs) instantiation of the Pippenger MSM kernels (msm_impl.hpp).
#include "msm_impl.hpp"
namespace zkmi {
template struct MsmEngine<Fq28>;
template hipError_t msm_build_table<Fq28>(const Affine<Fq28>*, uint64_t, const MsmPlan&, Affine<Fq28>**, hipStream_t);
template hipError_t bases_convert<Fq28>(const Affine<Fq>*, Affine<Fq28>*, uint64_t, hipStream_t);
template XYZZ<Fq> msm_combine_windows<Fq>(const XYZZ<Fq>*, int, int);
}  // namespace zkmi

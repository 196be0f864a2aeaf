// zkmi — BN254 G1 (F = BnFq28, 10 limbs) instantiation of the Pippenger MSM kernels (msm_impl.hpp):
// the MSM of a KZG commitment over the curve of the reference's own proving stack.
#include "msm_impl.hpp"
namespace zkmi {
template struct MsmEngine<BnFq28>;
template hipError_t msm_build_table<BnFq28>(const Affine<BnFq28>*, uint64_t, const MsmPlan&, Affine<BnFq28>**, hipStream_t);
template hipError_t bases_convert<BnFq28>(const Affine<BnFq>*, Affine<BnFq28>*, uint64_t, hipStream_t);
template XYZZ<BnFq> msm_combine_windows<BnFq>(const XYZZ<BnFq>*, int, int);
}  // namespace zkmi

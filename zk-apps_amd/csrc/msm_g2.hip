// zkmi — G2 (F = Fq2 over Fq28 limbs) instantiation of the Pippenger MSM kernels (msm_impl.hpp).
// Fully inlined field products: measured 12.5 ms vs 21 ms (out-of-line calls spill the
// operands to scratch) for the 2^20 accumulate; define ZK_CALL_MUL28 to get the call form.
#include "msm_impl.hpp"
namespace zkmi {
template struct MsmEngine<Fq2_28>;
template hipError_t msm_build_table<Fq2_28>(const Affine<Fq2_28>*, uint64_t, const MsmPlan&, Affine<Fq2_28>**, hipStream_t);
template hipError_t bases_convert<Fq2_28>(const Affine<Fq2>*, Affine<Fq2_28>*, uint64_t, hipStream_t);
template XYZZ<Fq2> msm_combine_windows<Fq2>(const XYZZ<Fq2>*, int, int);
}  // namespace zkmi

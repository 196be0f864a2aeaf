/* zkmi_testing.h — TEST SCAFFOLDING, not part of the product ABI.
 *
 * Everything declared here is compiled only with -DZKMI_TESTING and exported only by the A/B + testing library
 * zk-apps_amd/libzkmi_exp.so (`make -C zk-apps_amd/csrc experiments`); the product library libzkmi.so exports none of it
 * (tests/test_cpu_host.py::test_library_exports_every_declared_symbol checks both directions).  The objects these calls
 * create -- zkmi_bases_g1/g2, zkmi_bn_bases, zkmi_r1cs -- are the product's own types: a test creates a context with
 * libzkmi.so, manufactures its inputs here (the context handle is the same struct in both libraries, which are built from
 * one source tree), and hands them to the product's entry points.  The Python binding does exactly that (Zkmi.tlib).
 */
#ifndef ZKMI_TESTING_H
#define ZKMI_TESTING_H
#include "zkmi.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Synthetic bases P0 = G, P_{i+1} = P_i + [0xC0FFEE]G generated on the device: inputs of bench.py's MSM
 * legs and of the large-size property tests, without PCIe traffic. */
int32_t zkmi_bases_g1_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bases_g1** out);
int32_t zkmi_bases_g2_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bases_g2** out);
/* the slice P_first .. P_{first+n-1} of the same sequence (a rank's share of a point-split MSM) */
int32_t zkmi_bases_g1_synthetic_range(zkmi_ctx* ctx, uint64_t first, uint64_t n, zkmi_bases_g1** out);
/* BN254: P_i = [1 + i * 0xC0FFEE] G */
int32_t zkmi_bn254_bases_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bn_bases** out);

/* The hash-free chain stand-in of the first builds (Shielder-shaped: witness / public-input order of
 * UpdateNoteInput::new / update_note_circuit, update_note.rs:47-88, :121, :127; hashes replaced by a multiplication
 * chain): kept for the N = 128 golden fixture and as a fast relation of any size for property tests. */
int32_t zkmi_shielder_r1cs(uint32_t log_n, zkmi_r1cs** out);
int32_t zkmi_shielder_witness(uint32_t log_n, uint64_t seed, uint8_t* out_z /* 2^log_n x 32 B */);
int32_t zkmi_shielder_witness_from_input(uint32_t log_n, const zkmi_update_note_input* in, uint8_t* out_z);

/* Host-executed self-tests; *out_mismatches must be 0.
 *   fq28       the device limb representation (field28.hpp) against the 32-bit-limb host arithmetic
 *   assembly   the scalar multiplications of proof assembly (fixed-base tables, one- and two-point window forms, the
 *              shared inversion) against plain double-and-add in G1 and G2
 *   host_pool  the assembly pool under concurrent callers (every item exactly once)
 *   poseidon   the sparse partial-round form the kernels run against the plain 64-round definition */
int32_t zkmi_selftest_fq28(uint64_t seed, uint32_t iters, uint32_t* out_mismatches);
int32_t zkmi_selftest_assembly(uint64_t seed, uint32_t iters, uint32_t* out_mismatches);
int32_t zkmi_selftest_host_pool(uint32_t callers, uint32_t jobs, uint32_t* out_mismatches);
int32_t zkmi_selftest_poseidon(int32_t field, uint64_t seed, uint32_t iters, uint32_t* out_mismatches);
/* Device self-test of the quad-split complete addition (csrc/quad.hpp: one coordinate of an XYZZ point per lane of a quad)
 * against curve.hpp's one-lane addition on the device and the 32-bit-limb host arithmetic: n pairs with every special case
 * (o = a, o = -a, either at infinity, equal points in different representations), and 16-point sums over the quads of a wave. */
int32_t zkmi_selftest_quad_add(zkmi_ctx* ctx, uint64_t seed, uint32_t n, uint32_t* out_mismatches);
/* Test hook for the bucket set two MSMs share (the prover's L and H queries): sum_i a_i P_i + sum_i b_i P_i
 * with the first MSM's accumulation left unreduced and the second one's reduction taking both bucket arrays (prepared
 * bases run the shared-bucket schedule, others the windowed one).  Scalars in HBM. */
int32_t zkmi_selftest_msm_g1_sum2_dev(zkmi_ctx* ctx, const void* d_scalars_a, const void* d_scalars_b, uint64_t n,
                                      const zkmi_bases_g1* bases, uint8_t out_affine[96]);

#ifdef __cplusplus
}
#endif
#endif

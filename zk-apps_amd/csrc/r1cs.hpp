// zkmi — R1CS container (CSR, values in Montgomery form) and the
// Shielder-shaped synthetic relation.
#pragma once
#include <vector>
#include "../../include/zkmi.h"
#include "field.hpp"

struct zkmi_r1cs {
  uint32_t n_vars = 0, n_pub = 0, n_constraints = 0, log_n = 0;
  uint32_t tree_height = 0;  // update_note relations: Merkle height the shape was built for (0 = not such a relation)
  struct Csr {
    std::vector<uint32_t> rowptr, col;
    std::vector<zkmi::Fr> val;  // Montgomery form
  } m[3];                       // A, B, C
};

namespace zkmi {
void r1cs_finish_shape(zkmi_r1cs* r);
zkmi_r1cs* build_shielder_r1cs(uint32_t log_n);
void build_shielder_witness(uint32_t log_n, uint64_t seed, std::vector<Fr>* z_mont);
bool build_shielder_witness_from_input(uint32_t log_n, const zkmi_update_note_input& in, std::vector<Fr>* z_mont);
bool r1cs_satisfied(const zkmi_r1cs& r, const std::vector<Fr>& z_mont);
uint32_t pk_tree_height(const zkmi_pk* pk);  // groth16.hip: the tree height of the relation the key was made for, or 0
}  // namespace zkmi

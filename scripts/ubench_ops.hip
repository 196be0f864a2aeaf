// Issue cost of every VALU opcode in the loop body of k_accum_g1_nc: SIMD-ticks per wave-instruction
// at 1, 2 and 3 waves per SIMD (3 = the kernel's occupancy), independent chains, all 1024 SIMDs busy.
// scripts/isa_mix_report.py prices the kernel's static opcode mix with these figures.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_ops.hip -o scripts/_bin/ubench_ops && scripts/_bin/ubench_ops
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CHAINS 8
#define UNROLL 16

enum Op { MAD_I64_I32, MAD_U64_U32, AND_B32, ASHR_I64, LSHL_ADD_U64, MUL_LO_U32, SUB_U32, ASHR_I32, XAD_U32, LSHL_ADD_U32, LSHL_B32, ADD_U32,
          ALIGNBIT, LSHR_B64, OR3, CMP_EQ, MOV_B32, ADD_CO_PAIR, BFE_I32, OP_COUNT };
static const char* NAMES[] = {"v_mad_i64_i32", "v_mad_u64_u32", "v_and_b32", "v_ashrrev_i64", "v_lshl_add_u64", "v_mul_lo_u32", "v_sub_u32", "v_ashrrev_i32",
                              "v_xad_u32", "v_lshl_add_u32", "v_lshlrev_b32", "v_add_u32", "v_alignbit_b32", "v_lshrrev_b64", "v_or3_b32",
                              "v_cmp_eq_u32", "v_mov_b32", "v_add_co_u32+v_addc_co_u32", "v_bfe_i32"};

template <int OP>
__global__ void k(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  uint64_t acc[CHAINS];
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 7 + threadIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = seed + c + threadIdx.x;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        uint64_t carry;
        uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
        if (OP == MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=s"(carry) : "v"(a), "v"(b));
        if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=s"(carry) : "v"(a), "v"(b));
        if (OP == AND_B32) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ASHR_I64) asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(acc[c]));
        if (OP == LSHR_B64) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[c]));
        if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
        if (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == SUB_U32) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ASHR_I32) { asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(lo)); acc[c] = lo; }
        if (OP == LSHL_B32) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(lo)); acc[c] = lo; }
        if (OP == XAD_U32) { asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(a)); acc[c] = lo; }
        if (OP == LSHL_ADD_U32) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %1, %0, 28" : "+v"(lo) : "v"(hi)); acc[c] = lo | ((uint64_t)hi << 32); }
        if (OP == OR3) { asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(a)); acc[c] = lo; }
        if (OP == BFE_I32) { asm volatile("v_bfe_i32 %0, %0, 4, 20" : "+v"(lo)); acc[c] = lo; }
        if (OP == MOV_B32) { asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(hi)); acc[c] = lo | ((uint64_t)lo << 32); }
        if (OP == CMP_EQ) { asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(lo), "v"(b) : "vcc"); }
        if (OP == ADD_CO_PAIR) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc"); acc[c] = lo | ((uint64_t)hi << 32); }
      }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd) {
  const int cus = 256, iters = 400;
  const int threads = 64 * 4 * waves_per_simd;  // one block per CU
  const int nthreads = cus * threads;
  uint64_t *out, *cyc;
  hipMalloc(&out, 8 * nthreads);
  hipMalloc(&cyc, 8 * (nthreads / 64));
  hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), 0, 0, out, cyc, 10, 1u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), 0, 0, out, cyc, iters, 1u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(nthreads / 64);
  hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double per_wave = (double)iters * UNROLL * CHAINS * (OP == ADD_CO_PAIR ? 2 : 1);
  printf("OP %-28s waves/SIMD %d  SIMD-ticks/instr %6.2f  chip G wave-instr/s %8.1f\n", NAMES[OP], waves_per_simd, med / per_wave / waves_per_simd,
         per_wave * (nthreads / 64) / (ms * 1e-3) / 1e9);
  hipFree(out);
  hipFree(cyc);
}
template <int OP>
void sweep() {
  for (int w : {1, 2, 3}) run<OP>(w);
}
template <int OP>
void all() {
  sweep<OP>();
  if constexpr (OP + 1 < OP_COUNT) all<OP + 1>();
}
int main() {
  all<0>();
  return 0;
}

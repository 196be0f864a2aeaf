// Instruction-rate microbenchmark for gfx950 integer / fp64 VALU ops that a
// big-integer Montgomery multiplier can be built from.  Prints wave-cycles per
// wave-instruction per SIMD at 1..8 waves/SIMD (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CHAINS 8
#define UNROLL 16

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, ADD_U32, ADD_CO_PAIR, ADD3_U32, MAD_U64_DEP, FMA_F32, LSHL_ADD, MAD_U64_SGPR, OP_COUNT };
static const char* NAMES[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_fma_f64", "v_add_u32", "v_add_co+v_addc_co", "v_add3_u32", "v_mad_u64_u32(dep chain)", "v_fma_f32", "v_lshl_add_u32", "v_mad_u64_u32(sgpr multiplicand)"};

template <int OP>
__global__ void k(uint64_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  uint64_t acc[CHAINS];
  uint32_t a = seed + threadIdx.x, b = seed * 3 + 7 + threadIdx.x;
  double da = 1.000001 + threadIdx.x * 1e-9, db = 0.999999;
  float fa = 1.0001f, fb = 0.9999f;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = seed + c + threadIdx.x;
  double dacc[CHAINS];
  float facc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { dacc[c] = c + 1.0; facc[c] = c + 1.0f; }
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        uint64_t carry;
        uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
        if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=s"(carry) : "v"(a), "v"(b));
        if (OP == MAD_U64_SGPR) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=s"(carry) : "v"(a), "s"(seed));
        if (OP == MAD_U64_DEP) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[0]), "=s"(carry) : "v"(a), "v"(b));
        if (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == MUL_HI_U32) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == MAD_U32_U24) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(a)); acc[c] = lo; }
        if (OP == MUL_HI_U32_U24) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ADD3_U32) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(a)); acc[c] = lo; }
        if (OP == LSHL_ADD) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(b)); acc[c] = lo; }
        if (OP == ADD_CO_PAIR) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc"); acc[c] = lo | ((uint64_t)hi << 32); }
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dacc[c]) : "v"(db), "v"(da));
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(facc[c]) : "v"(fb), "v"(fa));
      }
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c] + (uint64_t)dacc[c] + (uint64_t)facc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd) {
  const int cus = 256, iters = 200;
  const int threads = 64 * 4 * waves_per_simd;  // per CU: one block per CU
  int nthreads = cus * threads;
  uint64_t *out, *cyc;
  hipMalloc(&out, 8 * nthreads);
  hipMalloc(&cyc, 8 * (nthreads / 64));
  int blk = threads > 1024 ? 1024 : threads;
  int grid = nthreads / blk;
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(blk), 0, 0, out, cyc, 10, 1u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(blk), 0, 0, out, cyc, iters, 1u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(nthreads / 64);
  hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  double med = (double)h[h.size() / 2];
  double per_wave_instr = (double)iters * UNROLL * CHAINS * (OP == ADD_CO_PAIR ? 2 : 1);
  // s_memtime ticks per wave-instruction for ONE wave; SIMD throughput = that / waves_per_simd
  double cyc_per_instr_wave = med / per_wave_instr;
  double total_instr = per_wave_instr * (nthreads / 64);
  printf("%-28s waves/SIMD=%d  ticks/instr/wave=%7.2f  SIMD-ticks/instr=%6.2f  chip Gwave-instr/s=%8.1f  (%.3f ms)\n", NAMES[OP], waves_per_simd,
         cyc_per_instr_wave, cyc_per_instr_wave / waves_per_simd, total_instr / (ms * 1e-3) / 1e9, ms);
  hipFree(out); hipFree(cyc);
}

template <int OP>
void sweep() { for (int w : {1, 2, 4, 8}) run<OP>(w); }

int main() {
  sweep<ADD_U32>(); sweep<FMA_F32>(); sweep<ADD3_U32>(); sweep<LSHL_ADD>(); sweep<ADD_CO_PAIR>();
  sweep<MAD_U64_U32>(); sweep<MAD_U64_SGPR>(); sweep<MAD_U64_DEP>(); sweep<MUL_LO_U32>(); sweep<MUL_HI_U32>();
  sweep<MAD_U32_U24>(); sweep<MUL_HI_U32_U24>(); sweep<FMA_F64>();
  return 0;
}

// What clock does the chip hold under which instruction mix?  ("name the limit behind 450 G/s")
//
// The accumulation kernels issue ~450 G wave-instructions/s at 1.85-2.0 GHz while k_build_table -- the same field
// arithmetic -- holds 2.35 GHz.  This sweep runs register-resident loops with a controlled share of 64-bit multiply-adds
// at a fixed occupancy (one-wave workgroups, W waves per SIMD on all 1024 SIMDs, ~1 s each) and reports, per body:
//   * the shader clock the waves themselves see: clock64() (s_memtime, core clock) over wall_clock64() (s_memrealtime,
//     100 MHz constant) between loop start and loop end;
//   * operations/s (the PMC run of the same binary, scripts/ubench_clock.sh, supplies SQ_INSTS_VALU per dispatch, from
//     which the issue rate follows);
//   * package power and temperature sampled by the host while the kernel runs (hwmon sysfs, rocm-smi as fallback).
// Bodies (all on Fq28 = 14 x 28-bit limbs, field28.hpp):
//   sqr      x <- x^2                      (105 + 196 multiply-adds of ~440 instructions)
//   mul      x <- x y                      (196 + 196 of ~541)
//   madd     XYZZ mixed addition, operand in registers (the accumulation loop without its gather)
//   madd_lds the same with the operand re-read from LDS every iteration (ds_read_b128 x 7)
//   add      x <- x + y, carry sweep       (no multiply-add at all)
//   mulK     one product followed by K additions (K = 2, 8, 32): multiply-add share 60 % .. 15 %
//   mad0/madr  a bare chain of v_mad_i64_i32 on all-zero / random operands (data-dependent power at a fixed mix)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-apps_amd/csrc scripts/ubench_clock.hip -o scripts/_bin/ubench_clock
// Usage: ubench_clock [seconds per body = 1.0] [waves per SIMD = 3]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "curve.hpp"
#include "field28.hpp"
#include "msm_impl.hpp"

using namespace zkmi;

enum Body { B_SQR, B_MUL, B_MADD, B_MADD_LDS, B_ADD, B_MUL2, B_MUL8, B_MUL32, B_MAD0, B_MADR, B_COUNT };
static const char* NAMES[B_COUNT] = {"sqr", "mul", "madd", "madd_lds", "add", "mul+2add", "mul+8add", "mul+32add", "mad_zero", "mad_random"};

__device__ __forceinline__ Fq28 seed_fq(uint32_t s) {
  Fq28 r;
#pragma unroll
  for (int i = 0; i < Fq28::NL; i++) {
    s = s * 1664525u + 1013904223u;
    r.l[i] = (int32_t)(s >> 4) & Fq28::MASK;
  }
  r.l[Fq28::NL - 1] &= 0xffff;
  return r;
}

struct Stamp {
  uint64_t core, real;
};

template <int BODY, int W>
__global__ void __launch_bounds__(64, W) k_body(uint64_t* out, Stamp* stamps, uint32_t iters, uint32_t seed) {
  __shared__ uint4 tile[7][64];
  const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
  uint64_t h = 0;
  uint64_t c0, r0, c1, r1;
  if constexpr (BODY == B_MADD || BODY == B_MADD_LDS) {
    XYZZ<Fq28> acc = {seed_fq(seed + gid), seed_fq(seed * 3 + gid + 7), Fq28::one(), Fq28::one()};
    Affine<Fq28> p = {seed_fq(seed + 11 * gid), seed_fq(seed + 13 * gid)};
    if constexpr (BODY == B_MADD_LDS) {
      const uint4* s = reinterpret_cast<const uint4*>(&p);
#pragma unroll
      for (int q = 0; q < 7; q++) tile[q][threadIdx.x] = s[q];
    }
    c0 = clock64(), r0 = wall_clock64();
    for (uint32_t it = 0; it < iters; it++) {
      if constexpr (BODY == B_MADD_LDS) {
        uint4* d = reinterpret_cast<uint4*>(&p);
#pragma unroll
        for (int q = 0; q < 7; q++) d[q] = tile[q][threadIdx.x];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      (void)madd_generic(acc, p, 0u);  // (random operands: the P = +-acc exit is never taken; its test runs)
    }
    c1 = clock64(), r1 = wall_clock64();
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) h += (uint32_t)acc.x.l[i] + (uint32_t)acc.y.l[i] + (uint32_t)acc.zz.l[i] + (uint32_t)acc.zzz.l[i];
  } else if constexpr (BODY == B_MAD0 || BODY == B_MADR) {
    // a dependent chain of 392 v_mad_i64_i32 per iteration (one product's worth); all-zero operands stay zero, random ones
    // keep toggling every bit
    const Fq28 x = seed_fq(seed + gid);
    // (the zeros are run-time values: iters < 2^31)
    int64_t m0 = BODY == B_MADR ? ((int64_t)x.l[0] << 20) ^ x.l[3] : (int64_t)(iters >> 31);
    const int32_t ma = BODY == B_MADR ? (x.l[1] | 1) : (int32_t)(iters >> 31);
    c0 = clock64(), r0 = wall_clock64();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
      for (int k = 0; k < 392; k++) m0 = (int64_t)ma * (int32_t)m0 + m0;
    }
    c1 = clock64(), r1 = wall_clock64();
    h = (uint64_t)m0;
  } else {
    Fq28 x = seed_fq(seed + gid), y = seed_fq(seed * 3 + gid + 7);
    c0 = clock64(), r0 = wall_clock64();
    for (uint32_t it = 0; it < iters; it++) {
      if constexpr (BODY == B_SQR) {
        x = x.sqr();
      } else if constexpr (BODY == B_MUL) {
        x = x * y;
      } else if constexpr (BODY == B_ADD) {
#pragma unroll
        for (int k = 0; k < 8; k++) x = x + y;
      } else {
        x = x * y;
        constexpr int K = BODY == B_MUL2 ? 2 : BODY == B_MUL8 ? 8 : 32;
#pragma unroll
        for (int k = 0; k < K; k++) y = y + x;
      }
    }
    c1 = clock64(), r1 = wall_clock64();
#pragma unroll
    for (int i = 0; i < Fq28::NL; i++) h += (uint32_t)x.l[i] + (uint32_t)y.l[i];
  }
  out[gid] = h;
  if (threadIdx.x == 0) stamps[blockIdx.x] = {c1 - c0, r1 - r0};
}

// ---- power / temperature sampling on the host while a kernel runs ----
static std::string find_hwmon(const char* leaf) {
  for (int card = 0; card < 16; card++)
    for (int hw = 0; hw < 16; hw++) {
      char path[256];
      snprintf(path, sizeof(path), "/sys/class/drm/card%d/device/hwmon/hwmon%d/%s", card, hw, leaf);
      FILE* f = fopen(path, "r");
      if (f) {
        fclose(f);
        return path;
      }
    }
  return "";
}
static double read_number(const std::string& path) {
  if (path.empty()) return -1;
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}
static double smi_power() {
  FILE* p = popen("rocm-smi --showpower --csv 2>/dev/null | tail -n +2 | head -1", "r");
  if (!p) return -1;
  char line[512] = {0};
  if (!fgets(line, sizeof(line), p)) line[0] = 0;
  pclose(p);
  const char* c = strrchr(line, ',');
  return c ? atof(c + 1) : -1;
}

template <int BODY, int W>
static void run(double seconds, uint64_t* d_out, Stamp* d_st, const std::string& pw_path, const std::string& tp_path) {
  const uint32_t waves = 1024u * W;
  // calibrate: iterations for ~`seconds`
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  uint32_t iters = 64;
  float ms = 0;
  for (int pass = 0; pass < 2; pass++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_body<BODY, W>), dim3(waves), dim3(64), 0, 0, d_out, d_st, iters, 12345u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    if (pass == 0) iters = (uint32_t)(iters * (seconds * 1e3 / (ms > 0.01f ? ms : 0.01f))) + 1;
  }
  // the measured launch: power sampled from a host thread every 50 ms
  std::atomic<bool> stop{false};
  std::vector<double> pw, tp;
  std::thread sampler([&] {
    while (!stop.load()) {
      double w = read_number(pw_path);
      if (w > 0) w /= 1e6;  // microwatts
      else w = smi_power();
      if (w > 0) pw.push_back(w);
      const double t = read_number(tp_path);
      if (t > 0) tp.push_back(t / 1e3);
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  });
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_body<BODY, W>), dim3(waves), dim3(64), 0, 0, d_out, d_st, iters, 999u);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  stop.store(true);
  sampler.join();
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> st(waves);
  hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
  double core = 0, real = 0;
  for (const Stamp& s : st) core += (double)s.core, real += (double)s.real;
  const double ghz = core / real * 0.1;  // wall_clock64 ticks at 100 MHz
  double pavg = 0, tmax = 0;
  // skip the first quarter of the samples (ramp)
  size_t from = pw.size() / 4, cnt = 0;
  for (size_t i = from; i < pw.size(); i++) pavg += pw[i], cnt++;
  if (cnt) pavg /= cnt;
  for (double t : tp) tmax = t > tmax ? t : tmax;
  const double ops_per_iter = BODY == B_ADD ? 8.0 : 1.0;
  printf("%-11s W=%d  iters %8u  %8.1f ms  sclk %.3f GHz  %9.3f G body-ops/s  %7.1f ns/op/wave  power %6.0f W (%zu samples)  temp %.0f C\n",
         NAMES[BODY], W, iters, ms, ghz, (double)waves * 64 * iters * ops_per_iter / (ms * 1e-3) / 1e9,
         ms * 1e6 / ((double)iters * ops_per_iter), pavg, cnt, tmax);
  fflush(stdout);
}

template <int W>
static void sweep(double seconds, uint64_t* d_out, Stamp* d_st, const std::string& pw, const std::string& tp) {
  run<B_SQR, W>(seconds, d_out, d_st, pw, tp);
  run<B_MUL, W>(seconds, d_out, d_st, pw, tp);
  run<B_MADD, W>(seconds, d_out, d_st, pw, tp);
  run<B_MADD_LDS, W>(seconds, d_out, d_st, pw, tp);
  run<B_MUL2, W>(seconds, d_out, d_st, pw, tp);
  run<B_MUL8, W>(seconds, d_out, d_st, pw, tp);
  run<B_MUL32, W>(seconds, d_out, d_st, pw, tp);
  run<B_ADD, W>(seconds, d_out, d_st, pw, tp);
  run<B_MAD0, W>(seconds, d_out, d_st, pw, tp);
  run<B_MADR, W>(seconds, d_out, d_st, pw, tp);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  const int w = argc > 2 ? atoi(argv[2]) : 3;
  uint64_t* d_out;
  Stamp* d_st;
  hipMalloc(&d_out, 8ull * 1024 * 4 * 64);
  hipMalloc(&d_st, sizeof(Stamp) * 1024 * 4);
  const std::string pw = find_hwmon("power1_average").empty() ? find_hwmon("power1_input") : find_hwmon("power1_average");
  const std::string tp = find_hwmon("temp1_input");
  printf("power source: %s; temperature: %s\n", pw.empty() ? "rocm-smi --showpower" : pw.c_str(), tp.empty() ? "-" : tp.c_str());
  if (w == 1) sweep<1>(seconds, d_out, d_st, pw, tp);
  else if (w == 2) sweep<2>(seconds, d_out, d_st, pw, tp);
  else sweep<3>(seconds, d_out, d_st, pw, tp);
  return 0;
}

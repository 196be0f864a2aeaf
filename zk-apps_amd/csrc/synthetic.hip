// zkmi — synthetic base points generated in HBM:
//   P_0 = G,  P_{i+1} = P_i + Q,  Q = [0xC0FFEE] G     (distinct points, same
//   recipe on G1 and on the twist), i.e. P_i = G + i*Q.
// Thread t owns the run i in [t*B, (t+1)*B): A_t = G + (t*B) Q by
// double-and-add, then P_{t*B+j} = A_t + (j Q) in affine coordinates with one
// shared inversion per run (Montgomery's trick).  Not on the proving path:
// it only manufactures bench / property-test inputs without PCIe traffic.
// Test scaffolding: compiled into libzkmi_exp.so only (-DZKMI_TESTING); the product library carries none of it.
#ifdef ZKMI_TESTING
#define ZK_CALL_MUL 1
#include <vector>
#include "curve.hpp"

namespace zkmi {

G1Affine g1_generator();
G2Affine g2_generator();

namespace {

constexpr int RUN = 64;

template <class T>
__device__ __forceinline__ T ld(const T* p) {
  T r;
  const uint4* s = reinterpret_cast<const uint4*>(p);
  uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
  return r;
}
template <class T>
__device__ __forceinline__ void st(T* p, const T& v) {
  const uint4* s = reinterpret_cast<const uint4*>(&v);
  uint4* d = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; i++) d[i] = s[i];
}

// table[j] = j*Q affine for j in [0, RUN) (table[0] unused), table[RUN] = Q
template <class F>
__global__ void __launch_bounds__(64)
k_synth(Affine<F>* __restrict__ out, const Affine<F>* __restrict__ table, Affine<F> g, uint64_t n, uint64_t first) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i0 = t * RUN;
  if (i0 >= n) return;
  // A = G + (first + i0) * Q: out[] holds the points P_first .. P_{first + n - 1}
  XYZZ<F> acc = XYZZ<F>::infinity();
  {
    const Affine<F> q = ld(table + RUN);
    const uint64_t k0 = first + i0;
    for (int b = 40; b >= 0; b--) {
      acc.dbl_inplace();
      if ((k0 >> b) & 1) acc.madd(q);
    }
    acc.madd(g);
  }
  const Affine<F> a = acc.to_affine();
  st(out + i0, a);
  const int cnt = (int)((n - i0 < RUN) ? (n - i0) : RUN);
  // forward: prefix products of d_j = x(jQ) - x(A), stored in out[i0+j].x
  F run = F::one();
  for (int j = 1; j < cnt; j++) {
    const Affine<F> qj = ld(table + j);
    st(&out[i0 + j].x, run);
    run = run * (qj.x - a.x);
  }
  F inv = run.inv();
  for (int j = cnt - 1; j >= 1; j--) {
    const Affine<F> qj = ld(table + j);
    const F d = qj.x - a.x;
    const F dinv = inv * ld(&out[i0 + j].x);  // 1/d_j
    inv = inv * d;
    const F lam = (qj.y - a.y) * dinv;
    const F x3 = lam.sqr() - a.x - qj.x;
    const F y3 = lam * (a.x - x3) - a.y;
    Affine<F> r = {x3, y3};
    st(out + i0 + j, r);
  }
}

template <class F>
hipError_t synth(Affine<F>* d_out, uint64_t n, const Affine<F>& g, hipStream_t stt, uint64_t first) {
  // host: Q = 0xC0FFEE * G and the run table
  uint32_t k[1] = {0xC0FFEEu};
  XYZZ<F> qx = scalar_mul(XYZZ<F>::from_affine(g), k, 1);
  Affine<F> q = qx.to_affine();
  std::vector<Affine<F>> table(RUN + 1);
  table[0] = Affine<F>::infinity();
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int j = 1; j < RUN; j++) {
    acc.madd(q);
    table[j] = acc.to_affine();
  }
  table[RUN] = q;
  Affine<F>* d_table = nullptr;
  hipError_t e = hipMalloc(&d_table, sizeof(Affine<F>) * table.size());
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(d_table, table.data(), sizeof(Affine<F>) * table.size(), hipMemcpyHostToDevice, stt);
  if (e == hipSuccess) {
    const uint64_t threads = (n + RUN - 1) / RUN;
    hipLaunchKernelGGL(k_synth<F>, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, stt, d_out, d_table, g, n, first);
    e = hipGetLastError();
  }
  hipError_t e2 = hipStreamSynchronize(stt);
  (void)hipFree(d_table);
  return e != hipSuccess ? e : e2;
}

}  // namespace

hipError_t synthetic_bases_g1(G1Affine* d_out, uint64_t n, hipStream_t stt, uint64_t first) {
  return synth<Fq>(d_out, n, g1_generator(), stt, first);
}
hipError_t synthetic_bases_g2(G2Affine* d_out, uint64_t n, hipStream_t stt, uint64_t first) {
  return synth<Fq2>(d_out, n, g2_generator(), stt, first);
}

}  // namespace zkmi

#endif  // ZKMI_TESTING

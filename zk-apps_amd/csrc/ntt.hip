// zkmi — radix-2 NTT over the BLS12-381 scalar field on gfx950.
//
// Reference locus: none in /root/reference.  Semantics
// = ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place and the coset
// variants (generator 7), as restated in oracle/ntt.py.
//
// Data layout: N Fr elements, 32 B each, Montgomery form, natural order in and
// out.  Twiddles w^k (k < N/2) and coset powers g^i live in HBM (precomputed
// per domain size, stay resident in the 256 MiB Infinity Cache).
//
// Kernel plan (LDS-staged butterflies): a transform is a bit-reversal copy
// followed by ceil(log N / S) decimation-in-time passes.  A pass owns S
// consecutive butterfly stages [t0, t0+S): every workgroup stages a tile of
// 2^S x 2^Q elements in LDS (2^S strided sub-problem points x 2^Q adjacent
// columns so that global loads stay >= 128 B contiguous), runs the S stages
// out of LDS with one barrier per stage, and writes the tile back.  With
// S = 10, Q = 2 a tile is 4096 x 32 B = 128 KiB of the CU's 160 KiB LDS, so
// N = 2^20 is exactly two passes over HBM (algorithmic minimum for a tile that
// must fit one CU).
#include "field.hpp"
#include "ntt.hpp"

namespace zkmi {

namespace {

struct alignas(16) FrV {
  uint4 lo, hi;
};

__device__ __forceinline__ Fr load_fr(const Fr* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void store_fr(Fr* p, const Fr& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ void k_bitrev_copy(const Fr* __restrict__ in, Fr* __restrict__ out, int log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = 1u << log_n;
  if (i >= n) return;
  uint32_t r = __brev(i) >> (32 - log_n);
  store_fr(out + r, load_fr(in + i));
}

// One DIT pass: stages [t0, t0+S) on an LDS tile of 2^S x 2^Q elements.
// tw[k] = w_N^k (forward) or w_N^-k (inverse), k < N/2.
// If scale != nullptr (last pass) every output is multiplied by scale[0]
// (N^-1 for the inverse) and, if post != nullptr, additionally by post[i]
// (coset inverse: g^-i).
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_ntt_dit_pass(Fr* __restrict__ data, const Fr* __restrict__ tw, int log_n, int t0, int S, int Q,
               const Fr* __restrict__ scale, const Fr* __restrict__ post) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  Fr* tile = reinterpret_cast<Fr*>(lds_raw);
  const int tile_log = S + Q;
  const uint32_t tile_n = 1u << tile_log;
  const uint32_t blk = blockIdx.x;

  // global index of tile element L:
  //   t0 == 0 : contiguous, g = blk * tile_n + L
  //   t0 >  0 : L = e * 2^Q + c ; g = hi << (t0+S) | e << t0 | mid << Q | c
  //             where blk = hi * 2^(t0-Q) + mid
  const uint32_t mid_bits = (t0 > 0) ? (uint32_t)(t0 - Q) : 0u;
  const uint32_t mid = blk & ((1u << mid_bits) - 1u);
  const uint32_t hi = blk >> mid_bits;
  auto gindex = [&](uint32_t L) -> uint32_t {
    if (t0 == 0) return blk * tile_n + L;
    uint32_t e = L >> Q, c = L & ((1u << Q) - 1u);
    return (hi << (t0 + S)) | (e << t0) | (mid << Q) | c;
  };

  for (uint32_t L = threadIdx.x; L < tile_n; L += THREADS) tile[L] = load_fr(data + gindex(L));
  __syncthreads();

  const uint32_t half = tile_n >> 1;
  for (int u = 0; u < S; u++) {
    const int t = t0 + u;  // global stage, butterfly distance 2^t
    // distance inside the tile
    const uint32_t dist_log = (t0 == 0) ? (uint32_t)u : (uint32_t)(u + Q);
    const uint32_t dist = 1u << dist_log;
    for (uint32_t b = threadIdx.x; b < half; b += THREADS) {
      // insert a zero bit at position dist_log
      uint32_t lo = b & (dist - 1u);
      uint32_t L0 = ((b >> dist_log) << (dist_log + 1)) | lo;
      uint32_t L1 = L0 | dist;
      uint32_t g0 = gindex(L0);
      uint32_t j = g0 & ((1u << t) - 1u);
      uint32_t k = j << (log_n - 1 - t);
      Fr w = load_fr(tw + k);
      Fr x = tile[L0];
      Fr y = tile[L1] * w;
      tile[L0] = x + y;
      tile[L1] = x - y;
    }
    __syncthreads();
  }

  for (uint32_t L = threadIdx.x; L < tile_n; L += THREADS) {
    uint32_t g = gindex(L);
    Fr v = tile[L];
    if (scale) v = v * load_fr(scale);
    if (post) v = v * load_fr(post + g);
    store_fr(data + g, v);
  }
}

__global__ void k_mul_pointwise(Fr* __restrict__ a, const Fr* __restrict__ b, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_fr(a + i, load_fr(a + i) * load_fr(b + i));
}

// table[i] = base^i * first, i < n, built by 2^12-element chunks:
// thread t of a chunk starts from base^(chunk*4096 + t*16) via square-and-multiply.
__global__ void k_power_table(Fr* __restrict__ out, Fr base, Fr first, uint32_t n) {
  uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
  if (i0 >= n) return;
  uint32_t e[1] = {i0};
  Fr v = base.pow(e, 1) * first;
  for (uint32_t k = 0; k < 16u && i0 + k < n; k++) {
    store_fr(out + i0 + k, v);
    v = v * base;
  }
}

__global__ void k_to_mont(Fr* __restrict__ a, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_fr(a + i, load_fr(a + i).to_mont());
}
__global__ void k_from_mont(Fr* __restrict__ a, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_fr(a + i, load_fr(a + i).from_mont());
}

}  // namespace

// ---------------------------------------------------------------------------
// host-side driver
// ---------------------------------------------------------------------------
static Fr host_fr_from_u64(uint64_t v) {
  Fr a = Fr::zero();
  a.l[0] = (uint32_t)v;
  a.l[1] = (uint32_t)(v >> 32);
  return a.to_mont();
}

Fr fr_root_of_unity(int log_n) {
  // 7^((r-1)/2^32), then square down to order 2^log_n
  static const uint32_t ROOT_2_32[8] = {0x439f0d2bu, 0x3829971fu, 0x8c2280b9u, 0xb6368350u,
                                        0x22c813b4u, 0xd09b6819u, 0xdfe81f20u, 0x16a2a19eu};
  Fr w;
  for (int i = 0; i < 8; i++) w.l[i] = ROOT_2_32[i];
  w = w.to_mont();
  for (int i = 32; i > log_n; i--) w = w.sqr();
  return w;
}

NttDomain::~NttDomain() {
  if (tw_fwd) (void)hipFree(tw_fwd);
  if (tw_inv) (void)hipFree(tw_inv);
  if (coset_fwd) (void)hipFree(coset_fwd);
  if (coset_inv) (void)hipFree(coset_inv);
  if (n_inv) (void)hipFree(n_inv);
  if (scratch) (void)hipFree(scratch);
}

hipError_t NttDomain::init(int log_n_, hipStream_t stream) {
  log_n = log_n_;
  const uint32_t n = 1u << log_n;
  const uint32_t half = n > 1 ? n / 2 : 1;
  hipError_t e;
  if ((e = hipMalloc(&tw_fwd, sizeof(Fr) * half)) != hipSuccess) return e;
  if ((e = hipMalloc(&tw_inv, sizeof(Fr) * half)) != hipSuccess) return e;
  if ((e = hipMalloc(&coset_fwd, sizeof(Fr) * n)) != hipSuccess) return e;
  if ((e = hipMalloc(&coset_inv, sizeof(Fr) * n)) != hipSuccess) return e;
  if ((e = hipMalloc(&n_inv, sizeof(Fr))) != hipSuccess) return e;
  if ((e = hipMalloc(&scratch, sizeof(Fr) * n)) != hipSuccess) return e;
  Fr w = fr_root_of_unity(log_n);
  Fr wi = w.inv();
  Fr g = host_fr_from_u64(7);
  Fr gi = g.inv();
  Fr ninv = host_fr_from_u64(n).inv();
  const int T = 256;
  auto blocks = [&](uint32_t cnt) { return (cnt + 16 * T - 1) / (16 * T); };
  hipLaunchKernelGGL(k_power_table, dim3(blocks(half)), dim3(T), 0, stream, tw_fwd, w, Fr::one(), half);
  hipLaunchKernelGGL(k_power_table, dim3(blocks(half)), dim3(T), 0, stream, tw_inv, wi, Fr::one(), half);
  hipLaunchKernelGGL(k_power_table, dim3(blocks(n)), dim3(T), 0, stream, coset_fwd, g, Fr::one(), n);
  hipLaunchKernelGGL(k_power_table, dim3(blocks(n)), dim3(T), 0, stream, coset_inv, gi, Fr::one(), n);
  if ((e = hipMemcpyAsync(n_inv, &ninv, sizeof(Fr), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  return hipGetLastError();
}

// In-place transform of d_data (Montgomery form, natural order).
hipError_t NttDomain::transform(Fr* d_data, bool inverse, bool coset, hipStream_t stream) {
  const uint32_t n = 1u << log_n;
  const int T = 256;
  if (log_n == 0) return hipSuccess;
  if (coset && !inverse)
    hipLaunchKernelGGL(k_mul_pointwise, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, coset_fwd, n);
  hipLaunchKernelGGL(k_bitrev_copy, dim3((n + T - 1) / T), dim3(T), 0, stream, d_data, scratch, log_n);
  // pass plan: S <= 10 stages per pass, Q = 2 adjacent columns for strided passes
  const Fr* tw = inverse ? tw_inv : tw_fwd;
  int t0 = 0;
  Fr* buf = scratch;
  while (t0 < log_n) {
    int S = log_n - t0;
    if (S > 10) S = 10;
    int Q = 0;
    if (t0 > 0) Q = (t0 >= 2) ? 2 : t0;
    // keep tile <= 4096 elements (128 KiB)
    while (S + Q > 12) S--;
    const bool last = (t0 + S == log_n);
    const uint32_t tile_n = 1u << (S + Q);
    const uint32_t nblk = n / tile_n;
    const size_t lds = (size_t)tile_n * sizeof(Fr);
    const Fr* scale = (last && inverse) ? n_inv : nullptr;
    const Fr* post = (last && inverse && coset) ? coset_inv : nullptr;
    if (tile_n >= 1024) {
      hipLaunchKernelGGL(k_ntt_dit_pass<1024>, dim3(nblk), dim3(1024), lds, stream, buf, tw, log_n, t0, S, Q,
                         scale, post);
    } else {
      hipLaunchKernelGGL(k_ntt_dit_pass<64>, dim3(nblk), dim3(64), lds, stream, buf, tw, log_n, t0, S, Q,
                         scale, post);
    }
    t0 += S;
  }
  hipError_t e = hipMemcpyAsync(d_data, scratch, sizeof(Fr) * n, hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

hipError_t ntt_to_mont(Fr* d, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_to_mont, dim3((n + 255) / 256), dim3(256), 0, s, d, n);
  return hipGetLastError();
}
hipError_t ntt_from_mont(Fr* d, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_from_mont, dim3((n + 255) / 256), dim3(256), 0, s, d, n);
  return hipGetLastError();
}

hipError_t ntt_enable_big_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_dit_pass<1024>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

}  // namespace zkmi

// zkmi — batched SHA-256 two-to-one hashing and the contract's Merkle tree on the device
// (SURVEY.md §8f-4).  The only part of the reference whose results are pinned by its own tests:
//   compute_hash / combine_merkle_hash = SHA-256(first.bytes || second.bytes)
//       shielder/contract/merkle.rs:24-28, shielder/mocked_zk/src/lib.rs:24-28
//   MerkleTree::add_leaf            merkle.rs:48-81 (a node that no inserted leaf has touched does
//                                   not exist and reads as Scalar 0, not as a hash of zeros)
//   test add_two_leaves_and_root    merkle.rs:115-132 (golden: tests/golden/mock_boundary.json)
// A message is exactly 64 bytes, so every hash is two compressions: the data block and a constant
// padding block whose message schedule is precomputed at compile time.
#include <string.h>
#include "ctx.hpp"

namespace zkmi {
namespace {

__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }
__device__ __forceinline__ uint32_t bswap(uint32_t x) { return __builtin_bswap32(x); }

// one compression; the 16-word schedule window is updated in place (w[i & 15])
__device__ __forceinline__ void compress(uint32_t h[8], uint32_t w[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
      const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i & 15];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    hh = g;
    g = f;
    f = e;
    e = d + t1;
    d = c;
    c = b;
    b = a;
    a = t1 + S0 + mj;
  }
  h[0] += a;
  h[1] += b;
  h[2] += c;
  h[3] += d;
  h[4] += e;
  h[5] += f;
  h[6] += g;
  h[7] += hh;
}

// out[i] = SHA-256(in[2 i] || in[2 i + 1]) for i < n; 32-byte elements, 16-byte vector accesses
__global__ __launch_bounds__(256) void k_sha256_pairs(const uint4* __restrict__ in, uint64_t n, uint4* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint4 v = in[4 * i + q];
    w[4 * q] = bswap(v.x);
    w[4 * q + 1] = bswap(v.y);
    w[4 * q + 2] = bswap(v.z);
    w[4 * q + 3] = bswap(v.w);
  }
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  compress(h, w);
  // padding block of a 64-byte message: 0x80, zeros, bit length 512 (constants fold through the unrolled schedule)
#pragma unroll
  for (int q = 0; q < 16; q++) w[q] = 0;
  w[0] = 0x80000000u;
  w[15] = 512u;
  compress(h, w);
  out[2 * i] = make_uint4(bswap(h[0]), bswap(h[1]), bswap(h[2]), bswap(h[3]));
  out[2 * i + 1] = make_uint4(bswap(h[4]), bswap(h[5]), bswap(h[6]), bswap(h[7]));
}

hipError_t sha_pairs(zkmi_ctx* ctx, const void* d_in, uint64_t n, void* d_out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_sha256_pairs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const uint4*>(d_in), n, static_cast<uint4*>(d_out));
  return hipGetLastError();
}

}  // namespace
}  // namespace zkmi

using namespace zkmi;

extern "C" {

int32_t zkmi_sha256_pairs_dev(zkmi_ctx* ctx, const void* d_in, uint64_t n_hashes, void* d_out) {
  ZK_ENTER(ctx);
  if (n_hashes && (!d_in || !d_out)) return ZKMI_ERR_BAD_ARG;
  if (ctx->timer()) ctx->timer()->begin(PH_MISC, ctx->stream);
  hipError_t e = sha_pairs(ctx, d_in, n_hashes, d_out);
  if (ctx->timer()) ctx->timer()->end(PH_MISC, ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "sha256 pairs");
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int32_t zkmi_sha256_pairs(zkmi_ctx* ctx, const uint8_t* in, uint64_t n_hashes, uint8_t* out) {
  ZK_ENTER(ctx);
  if (n_hashes && (!in || !out)) return ZKMI_ERR_BAD_ARG;
  ZK_HIP(ctx, ctx->staging(96 * n_hashes + 64));
  uint8_t* d_in = static_cast<uint8_t*>(ctx->d_tmp);
  uint8_t* d_out = d_in + 64 * n_hashes;
  if (n_hashes) ZK_HIP(ctx, hipMemcpyAsync(d_in, in, 64 * n_hashes, hipMemcpyHostToDevice, ctx->stream));
  hipError_t e = sha_pairs(ctx, d_in, n_hashes, d_out);
  if (e != hipSuccess) return ctx->hip_fail(e, "sha256 pairs");
  if (n_hashes) ZK_HIP(ctx, hipMemcpyAsync(out, d_out, 32 * n_hashes, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// The contract's tree after its first n_filled add_leaf calls, built level by level.
// d_nodes: 2^(log_leaves+1) - 1 elements of 32 B; the caller writes the first n_filled leaves; every
// other node is (re)written here: existing nodes get their hash, untouched ones are zero.
int32_t zkmi_sha256_merkle_tree_dev(zkmi_ctx* ctx, void* d_nodes, uint32_t log_leaves, uint64_t n_filled) {
  ZK_ENTER(ctx);
  if (!d_nodes || log_leaves > 30 || n_filled > (1ull << log_leaves)) return ZKMI_ERR_BAD_ARG;
  uint8_t* level = static_cast<uint8_t*>(d_nodes);
  const uint64_t n_leaves = 1ull << log_leaves;
  if (ctx->timer()) ctx->timer()->begin(PH_MISC, ctx->stream);
  // leaves that were never inserted read as zero, and so does everything above the leaf level for now
  ZK_HIP(ctx, hipMemsetAsync(level + 32 * n_filled, 0, 32 * (2 * n_leaves - 1 - n_filled), ctx->stream));
  uint64_t exist = n_filled;  // nodes of the current level some inserted leaf has touched
  for (uint32_t lv = 0; lv < log_leaves; lv++) {
    const uint64_t n_cur = n_leaves >> lv;
    uint8_t* next = level + 32 * n_cur;
    exist = (exist + 1) / 2;
    hipError_t e = sha_pairs(ctx, level, exist, next);
    if (e != hipSuccess) return ctx->hip_fail(e, "sha256 tree level");
    level = next;
  }
  if (ctx->timer()) ctx->timer()->end(PH_MISC, ctx->stream);
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

}  // extern "C"

// zkmi — context object behind the C ABI (include/zkmi.h): one HIP device, one
// stream, resident NTT domains and MSM workspaces.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/zkmi.h"
#ifdef ZKMI_TESTING
#include "../../include/zkmi_testing.h"  // test scaffolding: the A/B + testing library only
#endif
#include "curve.hpp"
#include "msm.hpp"
#include "ntt.hpp"

struct zkmi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_aux = nullptr;  // MSM reductions: overlap the next accumulation
  hipStream_t stream_aux2 = nullptr, stream_aux3 = nullptr;  // the prover spreads its five reductions over three streams
  hipStream_t stream_g2 = nullptr;   // G2 accumulation beside the G1 ones
  hipStream_t stream_front = nullptr;  // witness map + NTTs beside the MSMs over z
  hipStream_t stream_copy = nullptr;   // witness uploads / copies of the next proof
  hipStream_t stream_heavy = nullptr;  // ZKMI_HEAVY_ON=1 only: heavy-bucket kernels beside the accumulations; created at first use
  hipStream_t stream_sort = nullptr;   // the prover's digit sorts, beside the previous proof's accumulations
  hipStream_t stream_acc3 = nullptr;  // L accumulation of a single small proof (groth16.hip); created at first use
  hipStream_t stream_rz = nullptr;    // A/B library, ZKMI_RB1_STREAM=1: scaling and digit sort of r z; created at first use
  enum { PROOF_RING = 3 };  // proofs in flight in the batch prover (groth16.hip)
  hipEvent_t ev_sort[PROOF_RING] = {}, ev_z[PROOF_RING] = {}, ev_h[PROOF_RING] = {}, ev_sorth[PROOF_RING] = {};
  hipEvent_t ev_rz[PROOF_RING] = {};  // the digit sort of r z (B1 folded into the L + H reduction: groth16.hip)
  unsigned z_flip = 0;  // which of sort / sort_z2 the next z sort writes
  uint32_t group_override = 0;  // zkmi_ctx_set_group_size: proofs per group for keys created next (0 = automatic)
  // How the H MSM of the proof (group) in ring slot `par` runs: decided ONCE by prove_enqueue_z, consumed by prove_enqueue_h
  // and prove_finish (groth16.hip).
  //   H_OWN    its own sort, bucket set and reduction, queued by the second half
  //   H_INTO_L as H_OWN, but L + H share one reduction (slot of H; L has no result of its own): H's segment sums add L's bucket
  //            array to its own (A/B library, ZKMI_LH_MERGE=1: H's kernels accumulate INTO L's buckets)
  //   H_SORTED one small proof: sorted by the first half behind the transforms, accumulated by the second half
  //   H_FUSED  one small proof: sorted and accumulated by the first half, in one launch with A, B1 and L
  //   H_INTO_LB as H_INTO_L, and the B1 MSM -- taken over r z instead of z -- is a third source of that reduction (no result of its own)
  enum HMode { H_OWN = 0, H_INTO_L = 1, H_SORTED = 2, H_FUSED = 3, H_INTO_LB = 4 };
  int h_mode[PROOF_RING] = {H_OWN, H_OWN, H_OWN};
  std::string err;
  zkmi::PhaseTimer prof;
  std::map<int, std::unique_ptr<zkmi::NttDomain>> domains;
  zkmi::MsmSort sort;    // every MSM entry point; in the prover: the digit sort of z (A, B1, B2, L MSMs)
  zkmi::MsmSort sort_z2;  // second set of z-sort buffers: proof i+1 is sorted while proof i's accumulations read `sort`
  zkmi::MsmSort sort_h;  // the prover's digit sort of the h coefficients (own buffers: see prove_enqueue_h)
  zkmi::MsmSort sort_rz;  // digit sort of r z: B1 of a one-proof group, folded into the L + H reduction (reserved by such a key's setup)
  zkmi::MsmEngine<zkmi::Fq28> g1;
  zkmi::MsmEngine<zkmi::Fq2_28> g2;
  zkmi::MsmEngine<zkmi::BnFq28> g1_bn;  // BN254 G1 (bn254.hip)
  std::map<int, std::unique_ptr<zkmi::NttDomainBn>> domains_bn;
  void* d_tmp = nullptr;  // staging for host-buffer entry points
  uint64_t d_tmp_cap = 0;
  void* d_work = nullptr;  // limb-form work buffer of the NTT entry points
  void* d_pos[2] = {nullptr, nullptr};  // Poseidon constants per field (poseidon.hip)
  uint64_t d_work_cap = 0;

  // optional streams are created at first use: every HIP stream of a priority class shares a few hardware queues,
  // and streams that are never used only add aliasing (and slowed the traced bench down by orders of magnitude)
  hipError_t lazy_stream(hipStream_t* s, bool high_priority) {
    if (*s) return hipSuccess;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, high_priority ? hi : 0);
  }
  // every stream of the context, in creation order (lazily created ones included once they exist)
  std::vector<hipStream_t> all_streams() const {
    std::vector<hipStream_t> v;
    for (hipStream_t s : {stream, stream_aux, stream_aux2, stream_aux3, stream_g2, stream_front, stream_copy, stream_heavy,
                          stream_sort, stream_acc3, stream_rz})
      if (s) v.push_back(s);
    return v;
  }
  // wait for everything queued on the context (error paths hand the right to free buffers back to the caller)
  hipError_t drain() const {
    hipError_t first = hipSuccess;
    for (hipStream_t s : all_streams()) {
      const hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
  }
  zkmi::PhaseTimer* timer() { return prof.enabled ? &prof : nullptr; }
  int32_t fail(int32_t code, const std::string& msg) {
    err = msg;
    return code;
  }
  int32_t hip_fail(hipError_t e, const char* where) {
    err = std::string(where) + ": " + hipGetErrorString(e);
    return ZKMI_ERR_HIP;
  }
  hipError_t staging(uint64_t bytes) {
    if (bytes <= d_tmp_cap) return hipSuccess;
    if (d_tmp) (void)hipFree(d_tmp);
    d_tmp = nullptr;
    d_tmp_cap = 0;
    hipError_t e = hipMalloc(&d_tmp, bytes);
    if (e == hipSuccess) d_tmp_cap = bytes;
    return e;
  }
  zkmi::NttDomainBn* domain_bn(int log_n, hipError_t* e) {
    auto it = domains_bn.find(log_n);
    if (it != domains_bn.end()) {
      *e = hipSuccess;
      return it->second.get();
    }
    auto d = std::make_unique<zkmi::NttDomainBn>();
    *e = d->init(log_n, stream);
    if (*e != hipSuccess) return nullptr;
    auto* p = d.get();
    domains_bn[log_n] = std::move(d);
    return p;
  }
  zkmi::NttDomain* domain(int log_n, hipError_t* e) {
    auto it = domains.find(log_n);
    if (it != domains.end()) {
      *e = hipSuccess;
      return it->second.get();
    }
    auto d = std::make_unique<zkmi::NttDomain>();
    *e = d->init(log_n, stream);
    if (*e != hipSuccess) return nullptr;
    auto* p = d.get();
    domains[log_n] = std::move(d);
    return p;
  }
};

// d  : host representation (Montgomery R = 2^384), kept for read-back/export
// d28: device MSM representation (28-bit limbs, R = 2^392), what the kernels gather
struct zkmi_bases_g1 {
  zkmi_ctx* ctx;
  zkmi::G1Affine* d = nullptr;
  zkmi::Affine<zkmi::Fq28>* d28 = nullptr;
  zkmi::Affine<zkmi::Fq28>* tab = nullptr;  // optional: 2^(c w) * P_i for every digit position (zkmi_bases_g1_prepare)
  uint64_t n = 0;
};
struct zkmi_bases_g2 {
  zkmi_ctx* ctx;
  zkmi::G2Affine* d = nullptr;
  zkmi::Affine<zkmi::Fq2_28>* d28 = nullptr;
  zkmi::Affine<zkmi::Fq2_28>* tab = nullptr;  // optional table (zkmi_bases_g2_prepare)
  uint64_t n = 0;
};

#define ZK_HIP(ctx, call)                                   \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return (ctx)->hip_fail(_e, #call); \
  } while (0)

// first statement of every entry point that takes a ctx: bind the calling thread to the ctx's device
// (a process may hold one ctx per GPU; streams and allocations belong to their device)
#define ZK_ENTER(ctx)                                   \
  do {                                                  \
    if (!(ctx)) return ZKMI_ERR_BAD_ARG;                \
    ZK_HIP(ctx, hipSetDevice((ctx)->device));           \
  } while (0)

namespace zkmi {
// wire <-> internal conversions (host)
bool fr_from_wire(const uint8_t* b, Fr* out);  // canonical check, -> Montgomery
void fr_to_wire(const Fr& a, uint8_t* b);      // Montgomery -> canonical bytes
bool fq_from_wire(const uint8_t* b, Fq* out);
void fq_to_wire(const Fq& a, uint8_t* b);
bool g1_from_wire(const uint8_t* b, G1Affine* out, bool check_curve);
void g1_to_wire(const G1Affine& p, uint8_t* b);
bool g2_from_wire(const uint8_t* b, G2Affine* out, bool check_curve);
void g2_to_wire(const G2Affine& p, uint8_t* b);
bool fr_is_canonical(const uint8_t* b);
G1Affine g1_generator();
G2Affine g2_generator();
bool g1_on_curve(const G1Affine& p);
bool g2_on_curve(const G2Affine& p);
bool g1_in_subgroup(const G1Affine& p);  // [r]P = O
bool g2_in_subgroup(const G2Affine& p);
void g1_compress(const G1Affine& p, uint8_t out[48]);
bool g1_decompress(const uint8_t in[48], G1Affine* out);
void g2_compress(const G2Affine& p, uint8_t out[96]);
bool g2_decompress(const uint8_t in[96], G2Affine* out);
void g1_write_be(const G1Affine& p, uint8_t out[96]);   // uncompressed big-endian forms (arkworks Compress::No)
bool g1_read_be(const uint8_t in[96], G1Affine* out, bool check_curve);
void g2_write_be(const G2Affine& p, uint8_t out[192]);
bool g2_read_be(const uint8_t in[192], G2Affine* out, bool check_curve);
}  // namespace zkmi

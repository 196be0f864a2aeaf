// zkmi — C ABI entry points (include/zkmi.h): context, NTT, MSM, group helpers.
// Every function returns a status code; nothing throws across the boundary.
#include <string.h>
#include <new>
#include <type_traits>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "ctx.hpp"
#include "host_pool.hpp"
#include "msm_pipe.hpp"
#include "msm_impl.hpp"  // msm_combine_windows (host)

using namespace zkmi;

namespace zkmi {
#ifdef ZKMI_TESTING
hipError_t synthetic_bases_g1(G1Affine* d_out, uint64_t n, hipStream_t st, uint64_t first = 0);
hipError_t synthetic_bases_g2(G2Affine* d_out, uint64_t n, hipStream_t st, uint64_t first = 0);
#endif
}

template <class B>
static hipError_t bases_finish(B* b, hipStream_t st) {
  using F28 = typename std::remove_pointer<decltype(b->d28)>::type;
  hipError_t e = hipMalloc(&b->d28, sizeof(F28) * (b->n ? b->n : 1));
  if (e != hipSuccess) return e;
  if ((e = bases_convert(b->d, b->d28, b->n, st)) != hipSuccess) return e;
  return hipStreamSynchronize(st);
}
template <class B>
static void bases_destroy(B* b) {
  if (b->d) (void)hipFree(b->d);
  if (b->d28) (void)hipFree(b->d28);
  if (b->tab) (void)hipFree(b->tab);
  delete b;
}

template <class B, class A, bool (*FROM)(const uint8_t*, A*, bool), int W>
static int32_t bases_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, B** out) {
  if (!ctx || !out || (n && !affine) || n >= (1ull << 31)) return ZKMI_ERR_BAD_ARG;
  *out = nullptr;
  // wire -> internal on the host (canonical check), upload Montgomery form
  std::vector<A> h(n);
  for (uint64_t i = 0; i < n; i++)
    if (!FROM(affine + (uint64_t)W * i, &h[i], check != 0))
      return ctx->fail(ZKMI_ERR_NON_CANONICAL, "base point not canonical / not on curve");
  B* b = new (std::nothrow) B();
  if (!b) return ZKMI_ERR_BAD_ARG;
  b->ctx = ctx;
  b->n = n;
  hipError_t e = hipMalloc(&b->d, sizeof(A) * (n ? n : 1));
  if (e == hipSuccess && n) e = hipMemcpy(b->d, h.data(), sizeof(A) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = bases_finish(b, ctx->stream);
  if (e != hipSuccess) {
    bases_destroy(b);
    return ctx->hip_fail(e, "bases upload");
  }
  *out = b;
  return ZKMI_OK;
}

// Fixed bases used for many MSMs of their full length (an SRS, a proving-key query): build the table
// 2^(c w) * P_i once; zkmi_msm_g{1,2}[_dev] then run the shared-bucket schedule of the prover:
// ceil(255 / c) insertions per scalar with c up to 22 instead of 16 windows of 16 bits.
template <class F, class B>
static int32_t bases_prepare(zkmi_ctx* ctx, B* b) {
  if (!b || b->n == 0) return ZKMI_ERR_BAD_ARG;
  if (b->tab) return ZKMI_OK;
  const MsmPlan plan = msm_make_plan_shared(b->n);
  // table indices (digit * n + point) share a 32-bit word with the sign bit
  if ((uint64_t)plan.ndigits * b->n >= (1ull << 31)) return ctx->fail(ZKMI_ERR_BAD_ARG, "too many points for a table");
  hipError_t e = msm_build_table<F>(b->d28, b->n, plan, &b->tab, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    if (b->tab) (void)hipFree(b->tab);
    b->tab = nullptr;
    return ctx->hip_fail(e, "bases table");
  }
  return ZKMI_OK;
}


// sizeof of the handle types other translation units define (groth16.hip, r1cs.hip, bn254.hip)
uint64_t zkmi_layout_pk();
uint64_t zkmi_layout_r1cs();
uint64_t zkmi_layout_bn_bases();

extern "C" {

// "zkmi 0.1 (gfx950) src:<digest>[ exp]": the digest of the sources this binary was built from (Makefile: src_digest.o)
extern const char zkmi_src_digest_str[];
const char* zkmi_version(void) {
  static const std::string v = std::string("zkmi 0.1 (gfx950) src:") + zkmi_src_digest_str +
#ifdef ZKMI_EXPERIMENTS
                               " exp";
#else
                               "";
#endif
  return v.c_str();
}

// Sizes and member offsets of every struct that crosses between the product library and the A/B + testing library (tests
// create inputs in libzkmi_exp.so on a context made by libzkmi.so: zk-apps_amd/binding.py Zkmi.tlib compares the two
// fingerprints and refuses to pair libraries whose layouts differ -- a conditional member would otherwise corrupt memory
// silently).
int32_t zkmi_abi_layout_probe(uint64_t* out, uint32_t cap, uint32_t* out_n) {
  if (!out || !out_n) return ZKMI_ERR_BAD_ARG;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winvalid-offsetof"
  const uint64_t v[] = {
      sizeof(zkmi_ctx), offsetof(zkmi_ctx, prof), offsetof(zkmi_ctx, domains), offsetof(zkmi_ctx, sort), offsetof(zkmi_ctx, sort_rz),
      offsetof(zkmi_ctx, g1), offsetof(zkmi_ctx, g2), offsetof(zkmi_ctx, g1_bn), offsetof(zkmi_ctx, d_tmp), offsetof(zkmi_ctx, d_pos),
      sizeof(zkmi_bases_g1), sizeof(zkmi_bases_g2), sizeof(MsmPlan), sizeof(MsmSort), offsetof(MsmSort, plan), offsetof(MsmSort, readers),
      sizeof(MsmEngine<Fq28>), sizeof(MsmEngine<Fq2_28>), sizeof(MsmEngine<BnFq28>), offsetof(MsmEngine<Fq28>, slot_plan),
      offsetof(MsmEngine<Fq28>, cap_buckets), sizeof(PhaseTimer), sizeof(NttDomain), zkmi_layout_r1cs(), zkmi_layout_pk(),
      zkmi_layout_bn_bases(), sizeof(XYZZ<Fq28>), sizeof(Affine<Fq2_28>)};
#pragma clang diagnostic pop
  const uint32_t n = (uint32_t)(sizeof(v) / sizeof(v[0]));
  *out_n = n;
  if (cap < n) return ZKMI_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n; i++) out[i] = v[i];
  return ZKMI_OK;
}

// HIP version the library was compiled against and the one of the runtime it is bound to in this process (a Python
// process that also holds PyTorch-ROCm shares the wheel's runtime: zk-apps_amd/binding.py)
int32_t zkmi_hip_versions(int32_t* out_build, int32_t* out_runtime) {
  if (!out_build || !out_runtime) return ZKMI_ERR_BAD_ARG;
  *out_build = HIP_VERSION;
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return ZKMI_ERR_HIP;
  *out_runtime = v;
  return ZKMI_OK;
}

// The host side of the prover (host_pool.hpp): out[0] = CPUs' worth of time this process is granted (logical CPUs, affinity
// mask, cgroup quota), out[1] = processes of the job on this node (LOCAL_WORLD_SIZE), out[2] = threads the assembly of a
// group of proofs will occupy (the driving thread included), out[3] = worker threads the pool has started so far
int32_t zkmi_host_info(uint32_t out[4]) {
  if (!out) return ZKMI_ERR_BAD_ARG;
  out[0] = zkmi::host_cpus_granted();
  out[1] = zkmi::host_local_ranks();
  out[2] = zkmi::host_cpu_budget();
  out[3] = zkmi::HostPool::instance().workers();
  return ZKMI_OK;
}
const char* zkmi_host_info_string(void) {
  static thread_local std::string s;
  const zkmi::HostGrant& g = zkmi::host_grant();
  s = "cpus=" + std::to_string(g.cpus) + " (" + g.cpu_source + ") ranks=" + std::to_string(g.ranks) + " (" + g.rank_source +
      ") threads=" + std::to_string(zkmi::host_cpu_budget());
  return s.c_str();
}
// threads per process for proof assembly from now on (0 = back to the library's choice); a host that runs several
// contexts, or shares its CPUs with other work, knows better than the cgroup files
int32_t zkmi_set_host_threads(uint32_t n) {
  if (n > zkmi::HOST_THREADS_MAX) return ZKMI_ERR_BAD_ARG;
  zkmi::host_threads_override().store(n);
  return ZKMI_OK;
}

int32_t zkmi_device_count(int32_t* out_count) {
  if (!out_count) return ZKMI_ERR_BAD_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *out_count = (e == hipSuccess) ? n : 0;
  return (e == hipSuccess && n > 0) ? ZKMI_OK : ZKMI_ERR_NO_DEVICE;
}

// ZKMI_BACKTRACE=1: print the native stack on SIGSEGV / SIGABRT (debugging aid; the default handlers stay otherwise)
static void zkmi_crash_handler(int sig) {
  signal(SIGALRM, SIG_DFL);
  alarm(10);  // backtrace() may need a lock the crashed thread holds: never turn a crash into a hang
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[zkmi] fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

int32_t zkmi_ctx_create(int32_t device, zkmi_ctx** out_ctx) {
  if (!out_ctx) return ZKMI_ERR_BAD_ARG;
  *out_ctx = nullptr;
  if (getenv("ZKMI_BACKTRACE")) {
    void* warm[4];
    (void)backtrace(warm, 4);  // loads the unwinder now, not inside the signal handler
    signal(SIGSEGV, zkmi_crash_handler);
    signal(SIGABRT, zkmi_crash_handler);
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ZKMI_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return ZKMI_ERR_BAD_ARG;
  if (hipSetDevice(device) != hipSuccess) return ZKMI_ERR_HIP;
  zkmi_ctx* c = new (std::nothrow) zkmi_ctx();
  if (!c) return ZKMI_ERR_BAD_ARG;
  c->device = device;
  c->g1.nslots = 2;                     // MSMs by themselves use slots 0 and 1; a key setup asks for all twelve (ensure_slots)
  c->g2.nslots = zkmi_ctx::PROOF_RING;  // one G2 MSM per proof in flight
  c->g1_bn.nslots = 1;                  // BN254 MSMs run one at a time
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreate(&c->stream) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream_aux, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream_aux2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream_aux3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream_g2, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream_front, hipStreamNonBlocking,
                                  ZK_TUNE("ZKMI_FRONT_PRIO", 1) == 0 ? 0 : prio_hi) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking) != hipSuccess) {  // stream_sort / heavy / acc3: lazily
    delete c;
    return ZKMI_ERR_HIP;
  }
  for (int i = 0; i < zkmi_ctx::PROOF_RING; i++)
    if (hipEventCreateWithFlags(&c->ev_sort[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_z[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_h[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_sorth[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_rz[i], hipEventDisableTiming) != hipSuccess) {
      delete c;
      return ZKMI_ERR_HIP;
    }
  hipError_t e = ntt_enable_big_lds();
  if (e == hipSuccess) e = msm_sort_enable_big_lds();
  if (e != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return ZKMI_ERR_HIP;
  }
  *out_ctx = c;
  return ZKMI_OK;
}

int32_t zkmi_ctx_destroy(zkmi_ctx* ctx) {
  if (!ctx) return ZKMI_ERR_BAD_ARG;
  (void)hipSetDevice(ctx->device);
  (void)ctx->drain();
  ctx->domains.clear();
  ctx->domains_bn.clear();
  ctx->sort.release();
  ctx->sort_z2.release();
  ctx->sort_h.release();
  ctx->sort_rz.release();
  ctx->g1.release();
  ctx->g2.release();
  ctx->g1_bn.release();
  if (ctx->d_tmp) (void)hipFree(ctx->d_tmp);
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  for (void* p : ctx->d_pos)
    if (p) (void)hipFree(p);
  (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream_aux) (void)hipStreamDestroy(ctx->stream_aux);
  if (ctx->stream_aux2) (void)hipStreamDestroy(ctx->stream_aux2);
  if (ctx->stream_aux3) (void)hipStreamDestroy(ctx->stream_aux3);
  if (ctx->stream_g2) (void)hipStreamDestroy(ctx->stream_g2);
  for (int i = 0; i < zkmi_ctx::PROOF_RING; i++) {
    if (ctx->ev_sort[i]) (void)hipEventDestroy(ctx->ev_sort[i]);
    if (ctx->ev_z[i]) (void)hipEventDestroy(ctx->ev_z[i]);
    if (ctx->ev_h[i]) (void)hipEventDestroy(ctx->ev_h[i]);
    if (ctx->ev_sorth[i]) (void)hipEventDestroy(ctx->ev_sorth[i]);
    if (ctx->ev_rz[i]) (void)hipEventDestroy(ctx->ev_rz[i]);
  }
  if (ctx->stream_front) (void)hipStreamDestroy(ctx->stream_front);
  if (ctx->stream_heavy) (void)hipStreamDestroy(ctx->stream_heavy);
  if (ctx->stream_copy) (void)hipStreamDestroy(ctx->stream_copy);
  if (ctx->stream_sort) (void)hipStreamDestroy(ctx->stream_sort);
  if (ctx->stream_acc3) (void)hipStreamDestroy(ctx->stream_acc3);
  if (ctx->stream_rz) (void)hipStreamDestroy(ctx->stream_rz);
  delete ctx;
  return ZKMI_OK;
}

const char* zkmi_last_error(const zkmi_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int32_t zkmi_ctx_sync(zkmi_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return ZKMI_ERR_BAD_ARG;
  ZK_HIP(ctx, ctx->drain());
  ctx->prof.collect();
  return ZKMI_OK;
}

int32_t zkmi_prof_enable(zkmi_ctx* ctx, int32_t on) {
  ZK_ENTER(ctx);
  if (!ctx) return ZKMI_ERR_BAD_ARG;
  ctx->prof.enabled = on != 0;
  return ZKMI_OK;
}
int32_t zkmi_prof_reset(zkmi_ctx* ctx) {
  ZK_ENTER(ctx);
  if (!ctx) return ZKMI_ERR_BAD_ARG;
  ctx->prof.reset();
  return ZKMI_OK;
}
int32_t zkmi_prof_get(zkmi_ctx* ctx, int32_t phase, double* out_total_ms, uint64_t* out_launches) {
  ZK_ENTER(ctx);
  if (!ctx || phase < 0 || phase >= 16) return ZKMI_ERR_BAD_ARG;
  (void)ctx->drain();  // every stream that may carry timer events, the copy stream included
  ctx->prof.collect();
  if (out_total_ms) *out_total_ms = ctx->prof.total_ms[phase];
  if (out_launches) *out_launches = ctx->prof.count[phase];
  return ZKMI_OK;
}

// ---------------------------------------------------------------------------
// NTT
// ---------------------------------------------------------------------------
int32_t zkmi_ntt_fr_dev(zkmi_ctx* ctx, void* d_data, uint32_t log_n, int32_t inverse, int32_t coset) {
  ZK_ENTER(ctx);
  if (!ctx || !d_data || log_n > 26) return ZKMI_ERR_BAD_ARG;
  hipError_t e;
  NttDomain* dom = ctx->domain((int)log_n, &e);
  if (!dom) return ctx->hip_fail(e, "ntt domain init");
  const uint32_t n = 1u << log_n;
  // canonical words <-> limb form around the transform (not part of the timed phase)
  if (ctx->d_work_cap < (uint64_t)n * sizeof(Fr28)) {
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    ctx->d_work = nullptr;
    ctx->d_work_cap = 0;
    ZK_HIP(ctx, hipMalloc(&ctx->d_work, (uint64_t)n * sizeof(Fr28)));
    ctx->d_work_cap = (uint64_t)n * sizeof(Fr28);
  }
  Fr28* work = static_cast<Fr28*>(ctx->d_work);
  ZK_HIP(ctx, ntt_from_canonical(static_cast<const uint32_t*>(d_data), work, n, ctx->stream));
  PhaseTimer* t = ctx->timer();
  if (t) t->begin(PH_NTT, ctx->stream);
  e = dom->transform(work, inverse != 0, coset != 0, ctx->stream);
  if (t) t->end(PH_NTT, ctx->stream);
  if (e != hipSuccess) return ctx->hip_fail(e, "ntt transform");
  ZK_HIP(ctx, ntt_to_canonical(work, static_cast<uint32_t*>(d_data), n, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int32_t zkmi_ntt_fr(zkmi_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset) {
  ZK_ENTER(ctx);
  if (!ctx || !data || log_n > 26) return ZKMI_ERR_BAD_ARG;
  const uint64_t n = 1ull << log_n;
  for (uint64_t i = 0; i < n; i++)
    if (!fr_is_canonical(data + 32 * i)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "ntt input >= r");
  ZK_HIP(ctx, ctx->staging(n * 32));
  ZK_HIP(ctx, hipMemcpyAsync(ctx->d_tmp, data, n * 32, hipMemcpyHostToDevice, ctx->stream));
  int32_t rc = zkmi_ntt_fr_dev(ctx, ctx->d_tmp, log_n, inverse, coset);
  if (rc != ZKMI_OK) return rc;
  ZK_HIP(ctx, hipMemcpyAsync(data, ctx->d_tmp, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// ---------------------------------------------------------------------------
// bases
// ---------------------------------------------------------------------------
int32_t zkmi_bases_g1_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bases_g1** out) {
  ZK_ENTER(ctx);
  return bases_load<zkmi_bases_g1, G1Affine, g1_from_wire, 96>(ctx, affine, n, check, out);
}
int32_t zkmi_bases_g2_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bases_g2** out) {
  ZK_ENTER(ctx);
  return bases_load<zkmi_bases_g2, G2Affine, g2_from_wire, 192>(ctx, affine, n, check, out);
}
int32_t zkmi_bases_g1_free(zkmi_bases_g1* b) {
  if (!b) return ZKMI_ERR_BAD_ARG;
  bases_destroy(b);
  return ZKMI_OK;
}
int32_t zkmi_bases_g2_free(zkmi_bases_g2* b) {
  if (!b) return ZKMI_ERR_BAD_ARG;
  bases_destroy(b);
  return ZKMI_OK;
}

int32_t zkmi_msm_plan_query(uint64_t n, int32_t shared, uint32_t out[6]) {
  if (!out || n == 0 || n > MSM_MAX_TERMS) return ZKMI_ERR_BAD_ARG;
  const MsmPlan p = shared ? msm_make_plan_shared(n) : msm_make_plan(n);
  out[0] = (uint32_t)p.c;
  out[1] = (uint32_t)(shared ? p.ndigits : p.nwin);
  out[2] = (uint32_t)p.nwin;
  out[3] = p.nb;
  out[4] = (uint32_t)p.seg_log;
  out[5] = p.heavy_thr;
  return ZKMI_OK;
}

int32_t zkmi_bases_g1_prepare(zkmi_ctx* ctx, zkmi_bases_g1* b) {
  ZK_ENTER(ctx);
  return bases_prepare<Fq28>(ctx, b);
}
int32_t zkmi_bases_g2_prepare(zkmi_ctx* ctx, zkmi_bases_g2* b) {
  ZK_ENTER(ctx);
  return bases_prepare<Fq2_28>(ctx, b);
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_bases_g1_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bases_g1** out) {
  return zkmi_bases_g1_synthetic_range(ctx, 0, n, out);
}
#endif  // ZKMI_TESTING
#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_bases_g1_synthetic_range(zkmi_ctx* ctx, uint64_t first, uint64_t n, zkmi_bases_g1** out) {
  ZK_ENTER(ctx);
  if (!ctx || !out || n == 0 || n >= (1ull << 31) || first >= (1ull << 40)) return ZKMI_ERR_BAD_ARG;
  zkmi_bases_g1* b = new (std::nothrow) zkmi_bases_g1();
  if (!b) return ZKMI_ERR_BAD_ARG;
  b->ctx = ctx;
  b->n = n;
  hipError_t e = hipMalloc(&b->d, sizeof(G1Affine) * n);
  if (e == hipSuccess) e = synthetic_bases_g1(b->d, n, ctx->stream, first);
  if (e == hipSuccess) e = bases_finish(b, ctx->stream);
  if (e != hipSuccess) {
    bases_destroy(b);
    b = nullptr;
    return ctx->hip_fail(e, "synthetic g1 bases");
  }
  *out = b;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING
#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_bases_g2_synthetic(zkmi_ctx* ctx, uint64_t n, zkmi_bases_g2** out) {
  ZK_ENTER(ctx);
  if (!ctx || !out || n == 0 || n >= (1ull << 31)) return ZKMI_ERR_BAD_ARG;
  zkmi_bases_g2* b = new (std::nothrow) zkmi_bases_g2();
  if (!b) return ZKMI_ERR_BAD_ARG;
  b->ctx = ctx;
  b->n = n;
  hipError_t e = hipMalloc(&b->d, sizeof(G2Affine) * n);
  if (e == hipSuccess) e = synthetic_bases_g2(b->d, n, ctx->stream);
  if (e == hipSuccess) e = bases_finish(b, ctx->stream);
  if (e != hipSuccess) {
    bases_destroy(b);
    b = nullptr;
    return ctx->hip_fail(e, "synthetic g2 bases");
  }
  *out = b;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

int32_t zkmi_bases_g1_read(zkmi_ctx* ctx, const zkmi_bases_g1* b, uint64_t first, uint64_t count, uint8_t* out) {
  ZK_ENTER(ctx);
  if (!ctx || !b || !out || first + count > b->n) return ZKMI_ERR_BAD_ARG;
  std::vector<G1Affine> h(count);
  ZK_HIP(ctx, hipMemcpy(h.data(), b->d + first, sizeof(G1Affine) * count, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < count; i++) g1_to_wire(h[i], out + 96 * i);
  return ZKMI_OK;
}
int32_t zkmi_bases_g2_read(zkmi_ctx* ctx, const zkmi_bases_g2* b, uint64_t first, uint64_t count, uint8_t* out) {
  ZK_ENTER(ctx);
  if (!ctx || !b || !out || first + count > b->n) return ZKMI_ERR_BAD_ARG;
  std::vector<G2Affine> h(count);
  ZK_HIP(ctx, hipMemcpy(h.data(), b->d + first, sizeof(G2Affine) * count, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < count; i++) g2_to_wire(h[i], out + 192 * i);
  return ZKMI_OK;
}

// ---------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------
int32_t zkmi_msm_g1_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                        uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !d_scalars)) return ZKMI_ERR_BAD_ARG;
  const bool shared = bases->tab != nullptr && n == bases->n && n > 0;  // prepared bases, full length
  ZK_HIP(ctx, ctx->sort.reserve(n, shared));
  ZK_HIP(ctx, ctx->g1.reserve(n, shared));
  const MsmPlan wpl = msm_make_plan(n);
  if (!shared && msm_pipe_applies(wpl, n)) {
    // a big windowed MSM: the sort of the lower windows runs beside the accumulation of the upper ones (msm_pipe.hpp)
    ZK_HIP(ctx, ctx->sort_h.reserve(n));
    ZK_HIP(ctx, msm_pipe_enqueue(ctx, static_cast<const uint32_t*>(d_scalars), n, bases->d28, wpl));
    std::vector<G1XYZZ> win((size_t)wpl.nwin);
    ZK_HIP(ctx, msm_pipe_finish_windows(ctx, wpl, win.data()));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    g1_to_wire(msm_combine_windows<Fq>(win.data(), wpl.nwin, wpl.c).to_affine(), out_affine);
    return ZKMI_OK;
  }
  if (shared)
    ZK_HIP(ctx, ctx->sort.run_shared(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  else
    ZK_HIP(ctx, ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, shared ? bases->tab : bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(),
                                 PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  G1XYZZ res;
  ZK_HIP(ctx, ctx->g1.finish_host(&res));
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

int32_t zkmi_msm_g2_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g2* bases,
                        uint8_t out_affine[192]) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !d_scalars)) return ZKMI_ERR_BAD_ARG;
  const bool shared = bases->tab != nullptr && n == bases->n && n > 0;
  ZK_HIP(ctx, ctx->sort.reserve(n, shared));
  ZK_HIP(ctx, ctx->g2.reserve(n, shared));
  if (shared)
    ZK_HIP(ctx, ctx->sort.run_shared(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  else
    ZK_HIP(ctx, ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer()));
  ZK_HIP(ctx, ctx->g2.run_device(ctx->sort, shared ? bases->tab : bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(),
                                 PH_MSM_ACCUM_G2, PH_MSM_REDUCE_G2));
  G2XYZZ res;
  ZK_HIP(ctx, ctx->g2.finish_host(&res));
  g2_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

static int32_t upload_scalars(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n) {
  for (uint64_t i = 0; i < n; i++)
    if (!fr_is_canonical(scalars + 32 * i)) return ctx->fail(ZKMI_ERR_NON_CANONICAL, "scalar >= r");
  ZK_HIP(ctx, ctx->staging(n * 32 + 32));
  if (n) ZK_HIP(ctx, hipMemcpyAsync(ctx->d_tmp, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return ZKMI_OK;
}

int32_t zkmi_msm_g1(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bases_g1* bases,
                    uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !scalars)) return ZKMI_ERR_BAD_ARG;
  int32_t rc = upload_scalars(ctx, scalars, n);
  if (rc != ZKMI_OK) return rc;
  return zkmi_msm_g1_dev(ctx, ctx->d_tmp, n, bases, out_affine);
}
int32_t zkmi_msm_g2(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bases_g2* bases,
                    uint8_t out_affine[192]) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || (n && !scalars)) return ZKMI_ERR_BAD_ARG;
  int32_t rc = upload_scalars(ctx, scalars, n);
  if (rc != ZKMI_OK) return rc;
  return zkmi_msm_g2_dev(ctx, ctx->d_tmp, n, bases, out_affine);
}

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
int32_t zkmi_selftest_msm_g1_sum2_dev(zkmi_ctx* ctx, const void* d_scalars_a, const void* d_scalars_b, uint64_t n,
                                      const zkmi_bases_g1* bases, uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_affine || !d_scalars_a || !d_scalars_b || n == 0 || n > bases->n || n > MSM_MAX_TERMS) return ZKMI_ERR_BAD_ARG;
  const bool shared = bases->tab != nullptr && n == bases->n;
  ZK_HIP(ctx, ctx->sort.reserve(n, shared));
  ZK_HIP(ctx, ctx->sort_h.reserve(n, shared));
  ZK_HIP(ctx, ctx->g1.reserve(n, shared));
  const uint32_t* a = static_cast<const uint32_t*>(d_scalars_a);
  const uint32_t* b = static_cast<const uint32_t*>(d_scalars_b);
  if (shared) {
    ZK_HIP(ctx, ctx->sort.run_shared(a, n, ctx->stream, ctx->timer()));
    ZK_HIP(ctx, ctx->sort_h.run_shared(b, n, ctx->stream, ctx->timer()));
  } else {
    ZK_HIP(ctx, ctx->sort.run(a, n, ctx->stream, ctx->timer()));
    ZK_HIP(ctx, ctx->sort_h.run(b, n, ctx->stream, ctx->timer()));
  }
  const Affine<Fq28>* pts = shared ? bases->tab : bases->d28;
  // first MSM: slot 0, no reduction; second: slot 1, adds into slot 0's buckets and reduces both (same reduction stream)
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, pts, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, 0, nullptr,
                                 -1, MSM_RUN_NO_REDUCE));
  // (the product's mechanism: the second MSM's segment sums add both bucket arrays; A/B library with ZKMI_LH_MERGE=1: its
  // kernels add into the first one's array)
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort_h, pts, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, 1, nullptr, 0,
                                 ZK_TUNE("ZKMI_LH_MERGE", 2) == 1 ? 0 : MSM_RUN_ADD_AT_REDUCE));
  G1XYZZ res;
  ZK_HIP(ctx, ctx->g1.finish_host(&res, 1));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

int32_t zkmi_msm_g1_windows_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                uint64_t plan_n, uint8_t* out_windows_affine, uint32_t* out_nwin,
                                uint32_t* out_window_bits) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_windows_affine || !out_nwin || !out_window_bits || n > bases->n || n > MSM_MAX_TERMS ||
      plan_n > MSM_MAX_TERMS)
    return ZKMI_ERR_BAD_ARG;
  if (plan_n < n) plan_n = n;
  ZK_HIP(ctx, ctx->sort.reserve(plan_n));
  ZK_HIP(ctx, ctx->g1.reserve(plan_n));
  // every rank must use the same window width: plan from the global size
  MsmPlan pl = msm_make_plan(plan_n);
  ctx->sort.plan_override = pl.c;
  hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer());
  ctx->sort.plan_override = 0;
  if (e != hipSuccess) return ctx->hip_fail(e, "sort");
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  std::vector<G1XYZZ> win(ctx->sort.plan.nwin);
  ZK_HIP(ctx, ctx->g1.finish_host_windows(win.data()));
  for (int w = 0; w < ctx->sort.plan.nwin; w++) g1_to_wire(win[w].to_affine(), out_windows_affine + 96 * w);
  *out_nwin = (uint32_t)ctx->sort.plan.nwin;
  *out_window_bits = (uint32_t)ctx->sort.plan.c;
  return ZKMI_OK;
}

// The WINDOW split of BASELINE configs[3] as worded: this rank takes windows [w_first, w_first + w_count) of the plan of
// plan_n terms over ALL n points (it needs all scalars and all bases resident) and returns their sums; the ranks hold
// disjoint windows, so the exchange is a concatenation and the combination a Horner walk (zkmi_msm_g1_combine with
// infinity in the windows a rank does not own, or zkmi_msm_g1_window_split_allgather).
int32_t zkmi_msm_g1_window_range_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                     uint64_t plan_n, uint32_t w_first, uint32_t w_count, uint8_t* out_windows_affine,
                                     uint32_t* out_nwin_total, uint32_t* out_window_bits) {
  ZK_ENTER(ctx);
  if (!ctx || !bases || !out_windows_affine || n > bases->n || n > MSM_MAX_TERMS || plan_n > MSM_MAX_TERMS || (n && !d_scalars))
    return ZKMI_ERR_BAD_ARG;
  if (plan_n < n) plan_n = n;
  const MsmPlan pl = msm_make_plan(plan_n);
  if (w_count == 0 || w_first + w_count > (uint32_t)pl.nwin) return ZKMI_ERR_BAD_ARG;
  ZK_HIP(ctx, ctx->sort.reserve(plan_n));
  ZK_HIP(ctx, ctx->g1.reserve(plan_n));
  ctx->sort.plan_override = pl.c;
  ctx->sort.win_first = (int)w_first;
  ctx->sort.win_count = (int)w_count;
  const hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer());
  ctx->sort.plan_override = 0;
  ctx->sort.win_first = ctx->sort.win_count = 0;
  if (e != hipSuccess) return ctx->hip_fail(e, "sort");
  ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  std::vector<G1XYZZ> win(w_count);
  ZK_HIP(ctx, ctx->g1.finish_host_windows(win.data()));
  for (uint32_t w = 0; w < w_count; w++) g1_to_wire(win[w].to_affine(), out_windows_affine + 96 * w);
  if (out_nwin_total) *out_nwin_total = (uint32_t)pl.nwin;
  if (out_window_bits) *out_window_bits = (uint32_t)pl.c;
  return ZKMI_OK;
}

int32_t zkmi_msm_g1_multi(zkmi_ctx* const* ctxs, uint32_t n_dev, const void* const* d_scalars, const uint64_t* counts,
                          const zkmi_bases_g1* const* bases, uint8_t out_affine[96]) {
  if (!ctxs || !d_scalars || !counts || !bases || !out_affine || n_dev == 0 || n_dev > 64) return ZKMI_ERR_BAD_ARG;
  uint64_t total = 0;
  for (uint32_t d = 0; d < n_dev; d++) {
    if (!ctxs[d] || !bases[d] || counts[d] > bases[d]->n || (counts[d] && !d_scalars[d])) return ZKMI_ERR_BAD_ARG;
    total += counts[d];
  }
  if (total > MSM_MAX_TERMS) return ZKMI_ERR_BAD_ARG;
  // one window width for every slice, planned from the global size (as the multi-process path does)
  MsmPlan pl = msm_make_plan(total);
  // enqueue every device's sort + accumulation + reduction first, then collect: the devices run concurrently
  for (uint32_t d = 0; d < n_dev; d++) {
    zkmi_ctx* ctx = ctxs[d];
    ZK_ENTER(ctx);
    ZK_HIP(ctx, ctx->sort.reserve(total));  // sized for the global plan, like zkmi_msm_g1_windows_dev
    ZK_HIP(ctx, ctx->g1.reserve(total));
    ctx->sort.plan_override = pl.c;
    hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars[d]), counts[d], ctx->stream, ctx->timer());
    ctx->sort.plan_override = 0;
    if (e != hipSuccess) return ctx->hip_fail(e, "sort");
    ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases[d]->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1,
                                   PH_MSM_REDUCE_G1));
  }
  std::vector<G1XYZZ> sum(pl.nwin, G1XYZZ::infinity()), win(pl.nwin);
  for (uint32_t d = 0; d < n_dev; d++) {
    zkmi_ctx* ctx = ctxs[d];
    ZK_ENTER(ctx);
    if (ctx->sort.plan.nwin != pl.nwin) return ctx->fail(ZKMI_ERR_BAD_ARG, "window plan mismatch");
    ZK_HIP(ctx, ctx->g1.finish_host_windows(win.data()));
    for (int w = 0; w < pl.nwin; w++) sum[w].add(win[w]);
  }
  G1XYZZ res = msm_combine_windows<Fq>(sum.data(), pl.nwin, pl.c);
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

int32_t zkmi_msm_g1_combine(const uint8_t* windows_affine, uint32_t n_ranks, uint32_t nwin, uint32_t window_bits,
                            uint8_t out_affine[96]) {
  if (!windows_affine || !out_affine || nwin == 0 || nwin > 64 || window_bits == 0 || window_bits > 24)
    return ZKMI_ERR_BAD_ARG;
  std::vector<G1XYZZ> win(nwin, G1XYZZ::infinity());
  for (uint32_t r = 0; r < n_ranks; r++)
    for (uint32_t w = 0; w < nwin; w++) {
      G1Affine p;
      if (!g1_from_wire(windows_affine + 96ull * (r * nwin + w), &p, true)) return ZKMI_ERR_NON_CANONICAL;
      win[w].madd(p);
    }
  G1XYZZ total = msm_combine_windows<Fq>(win.data(), (int)nwin, (int)window_bits);
  g1_to_wire(total.to_affine(), out_affine);
  return ZKMI_OK;
}

// ---------------------------------------------------------------------------
// group helpers (host)
// ---------------------------------------------------------------------------
int32_t zkmi_g1_in_subgroup(const uint8_t affine[96]) {
  G1Affine p;
  if (!affine) return ZKMI_ERR_BAD_ARG;
  if (!g1_from_wire(affine, &p, true)) return ZKMI_ERR_NON_CANONICAL;
  return g1_in_subgroup(p) ? ZKMI_OK : ZKMI_ERR_NON_CANONICAL;
}
int32_t zkmi_g2_in_subgroup(const uint8_t affine[192]) {
  G2Affine p;
  if (!affine) return ZKMI_ERR_BAD_ARG;
  if (!g2_from_wire(affine, &p, true)) return ZKMI_ERR_NON_CANONICAL;
  return g2_in_subgroup(p) ? ZKMI_OK : ZKMI_ERR_NON_CANONICAL;
}
int32_t zkmi_g1_generator(uint8_t out[96]) {
  if (!out) return ZKMI_ERR_BAD_ARG;
  g1_to_wire(g1_generator(), out);
  return ZKMI_OK;
}
int32_t zkmi_g2_generator(uint8_t out[192]) {
  if (!out) return ZKMI_ERR_BAD_ARG;
  g2_to_wire(g2_generator(), out);
  return ZKMI_OK;
}
int32_t zkmi_g1_compress(const uint8_t affine[96], uint8_t out[48]) {
  G1Affine p;
  if (!affine || !out) return ZKMI_ERR_BAD_ARG;
  if (!g1_from_wire(affine, &p, true)) return ZKMI_ERR_NON_CANONICAL;
  g1_compress(p, out);
  return ZKMI_OK;
}
int32_t zkmi_g1_decompress(const uint8_t in[48], uint8_t out[96]) {
  G1Affine p;
  if (!in || !out) return ZKMI_ERR_BAD_ARG;
  if (!g1_decompress(in, &p)) return ZKMI_ERR_NON_CANONICAL;
  g1_to_wire(p, out);
  return ZKMI_OK;
}
int32_t zkmi_g2_compress(const uint8_t affine[192], uint8_t out[96]) {
  G2Affine p;
  if (!affine || !out) return ZKMI_ERR_BAD_ARG;
  if (!g2_from_wire(affine, &p, true)) return ZKMI_ERR_NON_CANONICAL;
  g2_compress(p, out);
  return ZKMI_OK;
}
int32_t zkmi_g2_decompress(const uint8_t in[96], uint8_t out[192]) {
  G2Affine p;
  if (!in || !out) return ZKMI_ERR_BAD_ARG;
  if (!g2_decompress(in, &p)) return ZKMI_ERR_NON_CANONICAL;
  g2_to_wire(p, out);
  return ZKMI_OK;
}
int32_t zkmi_g1_mul(const uint8_t affine[96], const uint8_t scalar[32], uint8_t out[96]) {
  G1Affine p;
  if (!affine || !scalar || !out) return ZKMI_ERR_BAD_ARG;
  if (!g1_from_wire(affine, &p, true) || !fr_is_canonical(scalar)) return ZKMI_ERR_NON_CANONICAL;
  uint32_t k[8];
  memcpy(k, scalar, 32);
  g1_to_wire(scalar_mul(G1XYZZ::from_affine(p), k, 8).to_affine(), out);
  return ZKMI_OK;
}
int32_t zkmi_g2_mul(const uint8_t affine[192], const uint8_t scalar[32], uint8_t out[192]) {
  G2Affine p;
  if (!affine || !scalar || !out) return ZKMI_ERR_BAD_ARG;
  if (!g2_from_wire(affine, &p, true) || !fr_is_canonical(scalar)) return ZKMI_ERR_NON_CANONICAL;
  uint32_t k[8];
  memcpy(k, scalar, 32);
  g2_to_wire(scalar_mul(G2XYZZ::from_affine(p), k, 8).to_affine(), out);
  return ZKMI_OK;
}
int32_t zkmi_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out[96]) {
  G1Affine p, q;
  if (!a || !b || !out) return ZKMI_ERR_BAD_ARG;
  if (!g1_from_wire(a, &p, true) || !g1_from_wire(b, &q, true)) return ZKMI_ERR_NON_CANONICAL;
  G1XYZZ s = G1XYZZ::from_affine(p);
  s.madd(q);
  g1_to_wire(s.to_affine(), out);
  return ZKMI_OK;
}
int32_t zkmi_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out[192]) {
  G2Affine p, q;
  if (!a || !b || !out) return ZKMI_ERR_BAD_ARG;
  if (!g2_from_wire(a, &p, true) || !g2_from_wire(b, &q, true)) return ZKMI_ERR_NON_CANONICAL;
  G2XYZZ s = G2XYZZ::from_affine(p);
  s.madd(q);
  g2_to_wire(s.to_affine(), out);
  return ZKMI_OK;
}

}  // extern "C"

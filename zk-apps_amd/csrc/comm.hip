// zkmi — the exchange step of a point-split MSM over RCCL, behind the C ABI (BASELINE config 3).
//
// One process per GPU; every rank runs the bucket method over its slice of the points with the window plan of the GLOBAL
// size and leaves per-(window, job) partial sums in HBM.  RCCL has no reduction over an elliptic-curve group law, so the
// "all-reduce" of BASELINE.json is an all-gather of the raw partials (ncclUint8, 16 x 13 x 192 B = 40 KB per rank at
// 2^26 terms) followed by the same combination on every rank: windows_from_partials per rank, one addition per window and
// rank, Horner over the windows.  The gather runs on the reduction stream, straight behind the tree sums, from the device
// buffer the reduction wrote -- no host round trip in front of the collective.
//
// libzkmi.so does NOT link RCCL: the five entry points it needs are resolved at first use, from an RCCL already in the
// process (the host's own: a communicator handed to zkmi_comm_from_nccl must belong to the library whose ncclAllGather is
// called on it) or else from librccl.so.1.  A host without RCCL can load the library and use everything but this file.
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <new>
#include <vector>
#include "ctx.hpp"
#include "msm_pipe.hpp"

namespace {

// the slice of rccl.h this file uses (the header is not needed at build time; values from rccl.h of ROCm 7.2:
// NCCL_UNIQUE_ID_BYTES = 128, ncclUint8 = 1, ncclSuccess = 0)
typedef void* nccl_comm_t;
struct nccl_unique_id {
  char internal[128];
};
typedef int (*fn_get_unique_id)(nccl_unique_id*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, nccl_unique_id, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
  bool ok = false;
  std::string why;
};

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    // an RCCL that is already mapped wins (dlopen by SONAME returns the loaded object; RTLD_NOLOAD first so that a process
    // that brought its own copy under another path is honoured through the global scope as well)
    // ZKMI_RCCL_LIB names the library file to use and nothing else is tried (a deployment whose RCCL lives outside the
    // loader's search path; the test-suite points it at a missing file -- ZKMI_ERR_RCCL -- and at an in-process all-gather
    // double that lets several ranks share one GPU, which RCCL itself refuses)
    // The variable puts a file into the prover's address space, so the product only takes what an administrator of this
    // account could have put there: an ABSOLUTE path to a regular file owned by root or by the effective user and not
    // writable by others (sshd's StrictModes rule); anything else is refused with ZKMI_ERR_RCCL -- never a silent fallback.
    void* h = nullptr;
    const char* forced = getenv("ZKMI_RCCL_LIB");
    if (forced && *forced) {
      struct stat sb;
      if (forced[0] != '/') {
        x.why = std::string("ZKMI_RCCL_LIB refused: not an absolute path: ") + forced;
        return x;
      }
      if (stat(forced, &sb) != 0) {
        x.why = std::string("RCCL not found: ZKMI_RCCL_LIB=") + forced + ": " + strerror(errno);
        return x;
      }
      if (!S_ISREG(sb.st_mode) || (sb.st_uid != 0 && sb.st_uid != geteuid()) || (sb.st_mode & S_IWOTH)) {
        x.why = std::string("ZKMI_RCCL_LIB refused: ") + forced + " must be a regular file owned by root or the effective user and not world-writable";
        return x;
      }
      h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
      if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather")) h = RTLD_DEFAULT;
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
      const char* de = dlerror();  // (one call: dlerror() clears the message it returns)
      x.why = std::string("RCCL not found: ") + (de ? de : (forced && *forced) ? forced : "librccl.so.1");
      return x;
    }
    x.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(h, "ncclGetUniqueId"));
    x.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(h, "ncclCommInitRank"));
    x.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(h, "ncclCommDestroy"));
    x.all_gather = reinterpret_cast<fn_all_gather>(dlsym(h, "ncclAllGather"));
    x.error_string = reinterpret_cast<fn_error_string>(dlsym(h, "ncclGetErrorString"));
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_gather;
    if (!x.ok) x.why = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
    return x;
  }();
  return r;
}

}  // namespace

struct zkmi_comm {
  zkmi_ctx* ctx = nullptr;   // identity check only after creation (the context may be destroyed first: `device` is what destroy uses)
  int device = 0;
  nccl_comm_t comm = nullptr;
  uint32_t n_ranks = 0, rank = 0;
  bool owned = false;         // created by zkmi_comm_init (destroyed with the handle) / borrowed from the host
  void* d_gather = nullptr;   // n_ranks x partial bytes, device
  uint64_t gather_cap = 0;
};

using namespace zkmi;

// room for the gathered slots (grows, never shrinks)
static hipError_t comm_reserve(zkmi_comm* comm, uint64_t bytes) {
  if (comm->gather_cap >= bytes) return hipSuccess;
  if (comm->d_gather) (void)hipFree(comm->d_gather);
  comm->d_gather = nullptr;
  comm->gather_cap = 0;
  const hipError_t e = hipMalloc(&comm->d_gather, bytes);
  if (e == hipSuccess) comm->gather_cap = bytes;
  return e;
}

// The combination every rank of a POINT split performs on the gathered array: rank k's slot holds the
// partials_per_msm(plan) sums its reduction wrote (window-major, jobs inside a window); per rank they become one sum per
// window, the ranks' windows are added, Horner over the windows.  Pure host arithmetic (also behind zkmi_msm_g1_combine_partials,
// which the CPU suite drives with many synthetic ranks).
static G1XYZZ combine_rank_partials(const MsmPlan& sp, const G1XYZZ* all, uint32_t n_ranks) {
  const int pts = MsmEngine<Fq28>::partials_per_msm(sp);
  std::vector<G1XYZZ> sum(sp.nwin, G1XYZZ::infinity()), win(sp.nwin);
  for (uint32_t k = 0; k < n_ranks; k++) {
    MsmEngine<Fq28>::windows_from_partials(sp, all + (size_t)k * pts, win.data());
    for (int w = 0; w < sp.nwin; w++) sum[w].add(win[w]);
  }
  return msm_combine_windows<Fq>(sum.data(), sp.nwin, sp.c);
}

// The combination of a WINDOW split: rank k's slot (slot_pts points) holds the partials of its windows
// [k nwin / R, (k + 1) nwin / R), whatever lies behind them in the slot is ignored.
static uint32_t split_first_window(uint32_t k, uint32_t nwin, uint32_t R) { return (uint32_t)(((uint64_t)k * nwin) / R); }
static MsmPlan split_sub_plan(const MsmPlan& pl, uint32_t k, uint32_t R) {
  MsmPlan q = pl;
  q.nwin_total = pl.nwin;
  q.win_first = (int)split_first_window(k, (uint32_t)pl.nwin, R);
  q.nwin = (int)(split_first_window(k + 1, (uint32_t)pl.nwin, R) - split_first_window(k, (uint32_t)pl.nwin, R));
  return q;
}
static uint32_t split_max_windows(const MsmPlan& pl, uint32_t R) {
  uint32_t max_w = 0;
  for (uint32_t k = 0; k < R; k++) max_w = std::max(max_w, (uint32_t)split_sub_plan(pl, k, R).nwin);
  return max_w;
}
static G1XYZZ combine_window_slots(const MsmPlan& pl, const G1XYZZ* all, uint64_t slot_pts, uint32_t R) {
  std::vector<G1XYZZ> win((size_t)pl.nwin, G1XYZZ::infinity());
  for (uint32_t k = 0; k < R; k++) {
    const MsmPlan q = split_sub_plan(pl, k, R);
    if (q.nwin > 0) MsmEngine<Fq28>::windows_from_partials(q, all + slot_pts * k, win.data() + q.win_first);
  }
  return msm_combine_windows<Fq>(win.data(), pl.nwin, pl.c);
}

// The combination of a 2-D split: R = P x Q ranks, rank k = g Q + q holds point group g (its slice of the points) and window
// range q of Q (split_sub_plan(pl, q, Q)); its slot holds the partials of those windows over those points.  A window's sum is
// the sum over the P point groups of the rank (g, range of the window)'s value.
static G1XYZZ combine_2d_slots(const MsmPlan& pl, const G1XYZZ* all, uint64_t slot_pts, uint32_t R, uint32_t Q) {
  std::vector<G1XYZZ> win((size_t)pl.nwin, G1XYZZ::infinity()), tmp((size_t)pl.nwin);
  for (uint32_t k = 0; k < R; k++) {
    const MsmPlan q = split_sub_plan(pl, k % Q, Q);
    if (q.nwin <= 0) continue;
    MsmEngine<Fq28>::windows_from_partials(q, all + slot_pts * k, tmp.data());
    for (int w = 0; w < q.nwin; w++) win[(size_t)q.win_first + w].add(tmp[w]);
  }
  return msm_combine_windows<Fq>(win.data(), pl.nwin, pl.c);
}

static int32_t rccl_fail(zkmi_ctx* ctx, int code, const char* where) {
  const Rccl& r = rccl();
  const std::string msg = std::string(where) + ": " + (r.error_string ? r.error_string(code) : "RCCL error");
  if (ctx) ctx->err = msg;
  return ZKMI_ERR_RCCL;
}

extern "C" {

int32_t zkmi_comm_unique_id(uint8_t out_id[128]) {
  if (!out_id) return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) {
    if (zkmi::debug_level()) fprintf(stderr, "zkmi: %s\n", r.why.c_str());  // (no context to carry the reason)
    return ZKMI_ERR_RCCL;
  }
  nccl_unique_id id;
  if (r.get_unique_id(&id) != 0) return ZKMI_ERR_RCCL;
  memcpy(out_id, id.internal, 128);
  return ZKMI_OK;
}

int32_t zkmi_comm_init(zkmi_ctx* ctx, uint32_t n_ranks, uint32_t rank, const uint8_t id[128], zkmi_comm** out) {
  ZK_ENTER(ctx);
  if (!id || !out || n_ranks == 0 || rank >= n_ranks) return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ctx->fail(ZKMI_ERR_RCCL, r.why);
  nccl_unique_id uid;
  memcpy(uid.internal, id, 128);
  nccl_comm_t c = nullptr;
  const int rc = r.comm_init_rank(&c, (int)n_ranks, uid, (int)rank);
  if (rc != 0) return rccl_fail(ctx, rc, "ncclCommInitRank");
  zkmi_comm* k = new (std::nothrow) zkmi_comm();
  if (!k) {
    (void)r.comm_destroy(c);
    return ZKMI_ERR_BAD_ARG;
  }
  k->ctx = ctx;
  k->device = ctx->device;
  k->comm = c;
  k->n_ranks = n_ranks;
  k->rank = rank;
  k->owned = true;
  *out = k;
  return ZKMI_OK;
}

int32_t zkmi_comm_from_nccl(zkmi_ctx* ctx, void* nccl_comm, uint32_t n_ranks, uint32_t rank, zkmi_comm** out) {
  ZK_ENTER(ctx);
  if (!nccl_comm || !out || n_ranks == 0 || rank >= n_ranks) return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ctx->fail(ZKMI_ERR_RCCL, r.why);
  zkmi_comm* k = new (std::nothrow) zkmi_comm();
  if (!k) return ZKMI_ERR_BAD_ARG;
  k->ctx = ctx;
  k->device = ctx->device;
  k->comm = nccl_comm;
  k->n_ranks = n_ranks;
  k->rank = rank;
  k->owned = false;
  *out = k;
  return ZKMI_OK;
}

int32_t zkmi_comm_destroy(zkmi_comm* comm) {
  if (!comm) return ZKMI_ERR_BAD_ARG;
  (void)hipSetDevice(comm->device);  // (not through comm->ctx: the context may already be gone)
  if (comm->d_gather) (void)hipFree(comm->d_gather);
  if (comm->owned && comm->comm && rccl().ok) (void)rccl().comm_destroy(comm->comm);
  delete comm;
  return ZKMI_OK;
}

// This rank's slice of a point-split MSM (n scalars at d_scalars against `bases`, window plan of plan_n = the global
// number of terms: every rank must pass the same), exchange, combination: out_affine receives the FULL result on every rank.
int32_t zkmi_msm_g1_allgather_combine(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n,
                                      const zkmi_bases_g1* bases, uint64_t plan_n, uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!comm || comm->ctx != ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || plan_n > MSM_MAX_TERMS ||
      (n && !d_scalars))
    return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ctx->fail(ZKMI_ERR_RCCL, r.why);
  if (plan_n < n) plan_n = n;
  // Everything that can fail on THIS rank alone -- arguments, the plan's fit, every allocation -- is settled before the
  // first launch: a rank that returned between its sort and the collective would leave its peers blocked in ncclAllGather.
  // (What can still fail below is a HIP or RCCL error: the communicator is unusable after one, see zkmi.h.)
  const MsmPlan pl = msm_make_plan(plan_n);  // one window width on every rank
  const int pts = MsmEngine<Fq28>::partials_per_msm(pl);
  if (pts > (int)MsmEngine<Fq28>::SLOT_PTS) return ctx->fail(ZKMI_ERR_BAD_ARG, "plan has more partial sums than a slot holds");
  const uint64_t bytes = sizeof(G1XYZZ) * (uint64_t)pts;
  ZK_HIP(ctx, ctx->sort.reserve(plan_n));
  ZK_HIP(ctx, ctx->g1.reserve(plan_n));
  if (msm_pipe_applies(pl, n)) {
    ZK_HIP(ctx, ctx->sort_h.reserve(plan_n));
    ZK_HIP(ctx, ctx->staging(bytes));
  }
  ZK_HIP(ctx, comm_reserve(comm, bytes * comm->n_ranks));
  std::vector<G1XYZZ> all((size_t)pts * comm->n_ranks);
  const void* send = ctx->g1.partial;
  MsmPlan sp = pl;
  if (msm_pipe_applies(pl, n)) {
    // two window groups, the second one's sort beside the first one's accumulation (msm_pipe.hpp); their partial sums lie in
    // slots 1 (windows [0, wb)) and 0 (the rest): brought together, in window order, in the staging buffer the gather sends
    const hipError_t e = msm_pipe_enqueue(ctx, static_cast<const uint32_t*>(d_scalars), n, bases->d28, pl);
    if (e != hipSuccess) return ctx->hip_fail(e, "pipelined sort / accumulation");
    const int per_window = pts / pl.nwin, wb = msm_pipe_split(pl);
    const size_t bytes_b = sizeof(G1XYZZ) * (size_t)per_window * wb;
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->g1.done[1], 0));  // (slot 0's reduction is on stream_aux itself)
    uint8_t* stage = static_cast<uint8_t*>(ctx->d_tmp);
    ZK_HIP(ctx, hipMemcpyAsync(stage, ctx->g1.partial + (size_t)MsmEngine<Fq28>::SLOT_PTS, bytes_b, hipMemcpyDeviceToDevice, ctx->stream_aux));
    ZK_HIP(ctx, hipMemcpyAsync(stage + bytes_b, ctx->g1.partial, (size_t)bytes - bytes_b, hipMemcpyDeviceToDevice, ctx->stream_aux));
    send = stage;
  } else {
    ctx->sort.plan_override = pl.c;
    const hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer());
    ctx->sort.plan_override = 0;
    if (e != hipSuccess) return ctx->hip_fail(e, "sort");
    ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
    // the reduction (slot 0, on stream_aux) has left partials_per_msm points in the device array `partial`
    sp = ctx->g1.slot_plan[0];
    if (MsmEngine<Fq28>::partials_per_msm(sp) != pts) return ctx->fail(ZKMI_ERR_BAD_ARG, "internal: the sort planned another window set");
  }
  const int rc = r.all_gather(send, comm->d_gather, (size_t)bytes, /* ncclUint8 */ 1, comm->comm, ctx->stream_aux);
  if (rc != 0) {
    (void)ctx->drain();
    return rccl_fail(ctx, rc, "ncclAllGather");
  }
  ZK_HIP(ctx, hipMemcpyAsync(all.data(), comm->d_gather, bytes * comm->n_ranks, hipMemcpyDeviceToHost, ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const G1XYZZ res = combine_rank_partials(sp, all.data(), comm->n_ranks);
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

// The window split as a collective: every rank holds ALL n scalars and bases and takes a contiguous share of the plan's
// windows (rank k: windows [k nwin / R, (k + 1) nwin / R) up to rounding; a rank beyond the window count takes none); the
// per-(window, job) partial sums are all-gathered in equal-sized slots and every rank walks the windows once.
int32_t zkmi_msm_g1_window_split_allgather(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n,
                                           const zkmi_bases_g1* bases, uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!comm || comm->ctx != ctx || !bases || !out_affine || n == 0 || n > bases->n || n > MSM_MAX_TERMS || !d_scalars) return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ctx->fail(ZKMI_ERR_RCCL, r.why);
  const MsmPlan pl = msm_make_plan(n);
  const uint32_t R = comm->n_ranks, nwin = (uint32_t)pl.nwin;
  const uint32_t w0 = split_first_window(comm->rank, nwin, R), w1 = split_first_window(comm->rank + 1, nwin, R);
  const uint32_t max_w = split_max_windows(pl, R);
  const int per_window = MsmEngine<Fq28>::partials_per_msm(pl) / pl.nwin;
  const uint64_t slot_pts = (uint64_t)per_window * max_w;
  const uint64_t slot_bytes = sizeof(G1XYZZ) * slot_pts;
  // (as in zkmi_msm_g1_allgather_combine: nothing between the first launch and the collective can fail on this rank alone)
  if (slot_pts > (uint64_t)MsmEngine<Fq28>::SLOT_PTS) return ctx->fail(ZKMI_ERR_BAD_ARG, "plan has more partial sums than a slot holds");
  ZK_HIP(ctx, ctx->sort.reserve(n));
  ZK_HIP(ctx, ctx->g1.reserve(n));
  ZK_HIP(ctx, comm_reserve(comm, slot_bytes * R));
  std::vector<G1XYZZ> all(slot_pts * R);
  if (w1 > w0) {
    ctx->sort.win_first = (int)w0;
    ctx->sort.win_count = (int)(w1 - w0);
    const hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer());
    ctx->sort.win_first = ctx->sort.win_count = 0;
    if (e != hipSuccess) return ctx->hip_fail(e, "sort");
    ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  }
  // (a rank's slot is the head of its `partial` array: whatever lies behind its own windows is ignored by the readers)
  const int rc = r.all_gather(ctx->g1.partial, comm->d_gather, (size_t)slot_bytes, /* ncclUint8 */ 1, comm->comm, ctx->stream_aux);
  if (rc != 0) {
    (void)ctx->drain();
    return rccl_fail(ctx, rc, "ncclAllGather");
  }
  ZK_HIP(ctx, hipMemcpyAsync(all.data(), comm->d_gather, slot_bytes * R, hipMemcpyDeviceToHost, ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const G1XYZZ res = combine_window_slots(pl, all.data(), slot_pts, R);
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

// The 2-D split as a collective (BASELINE configs[3] at 8 ranks: 4 point groups x 2 window ranges, or 2 x 4): rank
// k = g Q + q of R = P Q holds the n points of group g -- d_scalars / bases are ITS SLICE, as in the point split -- and sorts,
// accumulates and reduces only the windows of range q of the plan of plan_n terms: per rank 1 / R of the insertions like
// both 1-D splits, 1 / P of the points resident (the window split: all of them) and 1 / Q of the buckets to reduce (the
// point split: all of them -- the part of a rank's time that does not shrink with the slice).
// Exchange: the same all-gather of equal-sized slots, (partials per window) x (most windows any range owns) points each.
int32_t zkmi_msm_g1_split2d_allgather(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                      uint64_t plan_n, uint32_t window_groups, uint8_t out_affine[96]) {
  ZK_ENTER(ctx);
  if (!comm || comm->ctx != ctx || !bases || !out_affine || n > bases->n || n > MSM_MAX_TERMS || plan_n > MSM_MAX_TERMS || (n && !d_scalars) ||
      window_groups == 0 || comm->n_ranks % window_groups != 0)
    return ZKMI_ERR_BAD_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return ctx->fail(ZKMI_ERR_RCCL, r.why);
  if (plan_n < n) plan_n = n;
  const MsmPlan pl = msm_make_plan(plan_n);  // one window width on every rank
  const uint32_t R = comm->n_ranks, Q = window_groups, nwin = (uint32_t)pl.nwin;
  const uint32_t qi = comm->rank % Q;
  const uint32_t w0 = split_first_window(qi, nwin, Q), w1 = split_first_window(qi + 1, nwin, Q);
  const int per_window = MsmEngine<Fq28>::partials_per_msm(pl) / pl.nwin;
  const uint64_t slot_pts = (uint64_t)per_window * split_max_windows(pl, Q);
  const uint64_t slot_bytes = sizeof(G1XYZZ) * slot_pts;
  // (as in the 1-D collectives: nothing between the first launch and the all-gather can fail on this rank alone)
  if (slot_pts > (uint64_t)MsmEngine<Fq28>::SLOT_PTS) return ctx->fail(ZKMI_ERR_BAD_ARG, "plan has more partial sums than a slot holds");
  ZK_HIP(ctx, ctx->sort.reserve(plan_n));
  ZK_HIP(ctx, ctx->g1.reserve(plan_n));
  ZK_HIP(ctx, comm_reserve(comm, slot_bytes * R));
  std::vector<G1XYZZ> all(slot_pts * R);
  if (w1 > w0 && n) {
    ctx->sort.plan_override = pl.c;
    ctx->sort.win_first = (int)w0;
    ctx->sort.win_count = (int)(w1 - w0);
    const hipError_t e = ctx->sort.run(static_cast<const uint32_t*>(d_scalars), n, ctx->stream, ctx->timer());
    ctx->sort.plan_override = 0;
    ctx->sort.win_first = ctx->sort.win_count = 0;
    if (e != hipSuccess) return ctx->hip_fail(e, "sort");
    ZK_HIP(ctx, ctx->g1.run_device(ctx->sort, bases->d28, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1));
  } else {
    // nothing to add on this rank (more window ranges than windows, or an empty slice): its slot must still read as sums
    ZK_HIP(ctx, hipMemsetAsync(ctx->g1.partial, 0, (size_t)slot_bytes, ctx->stream_aux));
  }
  const int rc = r.all_gather(ctx->g1.partial, comm->d_gather, (size_t)slot_bytes, /* ncclUint8 */ 1, comm->comm, ctx->stream_aux);
  if (rc != 0) {
    (void)ctx->drain();
    return rccl_fail(ctx, rc, "ncclAllGather");
  }
  ZK_HIP(ctx, hipMemcpyAsync(all.data(), comm->d_gather, slot_bytes * R, hipMemcpyDeviceToHost, ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
  ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const G1XYZZ res = combine_2d_slots(pl, all.data(), slot_pts, R, Q);
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

// The two host combinations above on caller-supplied slots (no GPU, no RCCL): what every rank computes after the all-gather.
// partials = n_ranks slots of XYZZ points in the library's host form (4 x 48-byte Montgomery coordinates, the bytes the
// reductions leave in HBM).  window_split = 0: point split, a slot = partials_per_msm(plan of plan_n) points;
// window_split = 1: a slot = (partials per window) x (most windows any rank owns) points; window_split = Q >= 2: the 2-D
// split with Q window ranges (Q divides n_ranks), a slot = (partials per window) x (most windows any RANGE owns) points.
int32_t zkmi_msm_g1_combine_partials(const uint8_t* partials, uint32_t n_ranks, uint64_t plan_n, int32_t window_split,
                                     uint8_t out_affine[96]) {
  if (!partials || !out_affine || n_ranks == 0 || n_ranks > 4096 || plan_n == 0 || plan_n > MSM_MAX_TERMS) return ZKMI_ERR_BAD_ARG;
  const MsmPlan pl = msm_make_plan(plan_n);
  std::vector<G1XYZZ> all;
  G1XYZZ res;
  if (!window_split) {
    const size_t pts = (size_t)MsmEngine<Fq28>::partials_per_msm(pl);
    all.resize(pts * n_ranks);
    memcpy(all.data(), partials, sizeof(G1XYZZ) * all.size());
    res = combine_rank_partials(pl, all.data(), n_ranks);
  } else if (window_split == 1 || (uint32_t)window_split == n_ranks) {
    const uint64_t slot_pts = (uint64_t)(MsmEngine<Fq28>::partials_per_msm(pl) / pl.nwin) * split_max_windows(pl, n_ranks);
    all.resize(slot_pts * n_ranks);
    memcpy(all.data(), partials, sizeof(G1XYZZ) * all.size());
    res = combine_window_slots(pl, all.data(), slot_pts, n_ranks);
  } else {
    const uint32_t Q = (uint32_t)window_split;
    if (window_split < 0 || n_ranks % Q != 0) return ZKMI_ERR_BAD_ARG;
    const uint64_t slot_pts = (uint64_t)(MsmEngine<Fq28>::partials_per_msm(pl) / pl.nwin) * split_max_windows(pl, Q);
    all.resize(slot_pts * n_ranks);
    memcpy(all.data(), partials, sizeof(G1XYZZ) * all.size());
    res = combine_2d_slots(pl, all.data(), slot_pts, n_ranks, Q);
  }
  g1_to_wire(res.to_affine(), out_affine);
  return ZKMI_OK;
}

/* slot geometry of the two exchanges for a plan of plan_n terms: out[0] = windows, out[1] = partial sums per window,
 * out[2] = points per slot (point split), out[3] = points per slot (window split over n_ranks), out[4] = bytes per point,
 * out[5] = window bits, out[6] = log2 of the reduction's segment length, out[7] = top_spread_log */
int32_t zkmi_msm_exchange_layout(uint64_t plan_n, uint32_t n_ranks, uint32_t out[8]) {
  if (!out || plan_n == 0 || plan_n > MSM_MAX_TERMS || n_ranks == 0) return ZKMI_ERR_BAD_ARG;
  const MsmPlan pl = msm_make_plan(plan_n);
  const int pts = MsmEngine<Fq28>::partials_per_msm(pl);
  out[0] = (uint32_t)pl.nwin;
  out[1] = (uint32_t)(pts / pl.nwin);
  out[2] = (uint32_t)pts;
  out[3] = (uint32_t)(pts / pl.nwin) * split_max_windows(pl, n_ranks);
  out[4] = (uint32_t)sizeof(G1XYZZ);
  out[5] = (uint32_t)pl.c;
  out[6] = (uint32_t)pl.seg_log;
  out[7] = (uint32_t)pl.top_spread_log;
  return ZKMI_OK;
}

}  // extern "C"

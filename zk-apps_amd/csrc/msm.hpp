// zkmi — host-side handles for the Pippenger MSM pipeline (see msm_impl.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "curve.hpp"
#include "field28.hpp"
#include "tune.hpp"

namespace zkmi {

// hipEvent-based per-phase timers on the launch stream (bench.py reads these
// through zkmi_prof_get; SURVEY.md §8d "hipEvent around the kernel sequence").
struct PhaseTimer {
  enum { MAX_PENDING = 4096 };
  hipEvent_t ev0[MAX_PENDING], ev1[MAX_PENDING];
  int phase_of[MAX_PENDING];
  int n_pending = 0, n_created = 0;
  bool enabled = false;
  double total_ms[16] = {0};
  uint64_t count[16] = {0};
  ~PhaseTimer();
  void begin(int phase, hipStream_t st);
  void end(int phase, hipStream_t st);
  void collect();  // call after the stream is synchronised
  void reset();
};

enum Phase {
  PH_MSM_SORT = 0,
  PH_MSM_ACCUM_G1 = 1,
  PH_MSM_REDUCE_G1 = 2,
  PH_MSM_ACCUM_G2 = 3,
  PH_MSM_REDUCE_G2 = 4,
  PH_NTT = 5,
  PH_WITNESS = 6,
  PH_MISC = 7,
  PH_COUNT = 8
};

// The sort addresses sorted[] with 32-bit slots: windowed mode uses regions w * n (nwin = 16 at c = 16)
// and shared mode packs (digit * n + point) below the sign bit, so both need 16 n < 2^32.
constexpr uint64_t MSM_MAX_TERMS = (1ull << 28) - 1;

constexpr int MSM_BIG_WINDOW = 20;  // window bits of the windowed schedule from 2^24 terms on (msm_sort.hip pick_window)
constexpr int MSM_MAX_WINDOW = 22;  // widest window the partitioned sort supports (16 windows x 16.. partitions <= 256 groups)

struct MsmPlan {
  int c = 0;        // digit bits (signed digits)
  int nwin = 0;     // windowed: number of windows;  shared: number of 2^15-bucket partitions
  uint32_t nb = 0;  // windowed: buckets per window = 2^(c-1);  shared: buckets per partition
  uint64_t n = 0;
  // shared-bucket mode (precomputed tables 2^(c w) P_i, see msm_impl.hpp): all
  // ndigits digits of a scalar land in ONE bucket set of 2^(c-1) = nwin * nb buckets
  bool shared = false;
  int ndigits = 0;
  // buckets above heavy_thr points (>= 8x the mean load, >= 256) are reduced by whole
  // workgroups; load ordering uses the key (count >> heavy_shift) <= 256
  uint32_t heavy_thr = 256;
  int heavy_shift = 0;
  // k_segreduce walks 2^seg_log buckets per thread (a dependent chain of 2 * 2^seg_log additions):
  // 16 buckets when the chip is full anyway, 4 for small problems whose reduction is pure latency
  int seg_log = 4;
  // batched shared sort (MsmSort::run_shared_batch): nwin = vectors x vec_parts, vector v owns partitions
  // [v * vec_parts, (v + 1) * vec_parts)
  int vec_parts = 1;
  // windowed plans with more than 2^15 buckets per window (c > 16, MsmSort::run_windowed_big): the partial TOP window only
  // has digits below 2^15 -- every scalar would land in one of its 2^(c-1-15) partitions, in buckets 30 x the mean load.
  // Its entries are spread over all 2^top_spread_log partitions of the window by point index instead (bucket = (point mod
  // 2^top_spread_log) * 2^15 + |digit| - 1); the reduction weights a bucket by its index + 1 = partition * 2^15 + |digit|,
  // and the partition's share of that weight sits in the top top_spread_log bits of the SEGMENT index, whose bit sums
  // the host simply leaves out for this window (MsmEngine::finish_host_windows).
  int top_spread_log = 0;
  // a sort restricted to the windows [win_first, win_first + nwin) of a windowed plan with nwin_total windows (a rank of a
  // WINDOW-split MSM: BASELINE configs[3] as worded, MsmSort::win_first / win_count); nwin_total = 0: all windows
  int win_first = 0, nwin_total = 0;
  int total_windows() const { return nwin_total ? nwin_total : nwin; }
};
MsmPlan msm_make_plan(uint64_t n);
MsmPlan msm_make_plan_c(uint64_t n, int c);
MsmPlan msm_make_plan_shared(uint64_t n);
MsmPlan msm_make_plan_shared_batch(uint64_t n, uint32_t batch);  // the plan MsmSort::run_shared_batch(n, .., batch) will use

// Bucket scatter: signed-digit decomposition + counting sort of point indices
// by (window, bucket).  Shared by every MSM over the same scalar vector.
struct MsmSort {
  uint32_t* count = nullptr;      // nwin*nb   points per bucket
  uint32_t* begin = nullptr;      // nwin*nb   first slot of the bucket in sorted[] (window w owns [w*n, (w+1)*n))
  uint32_t* blockhist = nullptr;  // nwin*nch*nb  per-(window, chunk) tile histogram -> tile base slots
  uint32_t* perm = nullptr;       // nwin*nb   bucket ids of ALL windows ordered by descending load
  uint32_t* heavy = nullptr;      // [0] = number of heavy buckets, [1..] their ids
  uint32_t* order_bins = nullptr; // load-ordering: global key histogram -> running offsets
  uint32_t* part_total = nullptr; // shared mode: entries per partition
  uint32_t* blkcnt = nullptr;     // shared mode record pre-pass: per-(block, partition) counts -> slots
  uint32_t* fpart = nullptr;      // fine-partition sort: [q] first record slot, [NP + q] records of fine partition q
  uint32_t* rec_entry = nullptr;  // records grouped by partition: table index | sign
  uint32_t* rec_bkt = nullptr;    //                               bucket id inside the partition
  uint32_t* rec_aux = nullptr;    // big windowed plans: bucket ids of the first-level records (msm_sort.hip k_rec_split)
  uint32_t* big_ws = nullptr;     // ... and the list + chunk counters of oversized fine partitions (k_big_*)
  uint32_t* sorted = nullptr;     // nwin*n    point index | sign<<31
  uint64_t cap_entries = 0, cap_buckets = 0, cap_hist = 0;
  MsmPlan plan;
  int plan_override = 0;  // force window bits (multi-GPU split: all ranks must agree)
  int win_first = 0, win_count = 0;  // run(): only these windows of the plan (win_count = 0: all); reset by the caller
  // run_windowed_big: log2 of the buckets per LDS-histogram tile -- 15 (128 KB of counters: a CU to itself) or 14 (64 KB: fits
  // beside an occupancy-capped accumulation, MSM_RUN_TWO_WAVES); 14 only for window ranges WITHOUT the plan's top window
  int big_nb_log = 15;
  // events of kernels on OTHER streams that still read count/begin/heavy/sorted (the heavy-bucket
  // kernels of MsmEngine::run_device): the next sort waits for them before it overwrites the buffers
  mutable std::vector<hipEvent_t> readers;
  hipError_t wait_readers(hipStream_t st);
  ~MsmSort() { release(); }
  void release();
  bool has_shared = false;  // buffers sized for the shared-bucket plan too
  hipError_t reserve(uint64_t n, bool shared_too = false);
  hipError_t allocate(uint64_t ne, uint64_t nbk, uint64_t nh, bool shared);
  hipError_t run(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof);
  hipError_t run_windowed_big(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof);  // c > 16, called by run()
  // shared-bucket mode: sorted[] entries are table indices (digit * n + point) | sign << 31
  hipError_t run_shared(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof);
  // `batch` scalar vectors over the same bases, one bucket set each (small domains: msm_sort.hip)
  hipError_t run_shared_batch(const uint32_t* d_scalars, uint64_t n, uint64_t stride_words, uint32_t batch, hipStream_t st,
                              PhaseTimer* prof);
  hipError_t reserve_batch(uint64_t n, uint32_t batch);
};

hipError_t msm_sort_enable_big_lds();  // per device, see msm_sort.hip

// device field F (lazily reduced 28-bit limbs) <-> host field (32-bit limbs)
template <class F> struct HostFieldOf;
template <> struct HostFieldOf<Fq28> { using type = Fq; };
template <> struct HostFieldOf<Fq2_28> { using type = Fq2; };
template <> struct HostFieldOf<BnFq28> { using type = BnFq; };

// MsmEngine::run_device flags (msm_impl.hpp run_device_multi); MSM_RUN_TWO_WAVES: the G1 accumulation capped at two waves per SIMD
enum { MSM_RUN_NO_REDUCE = 1, MSM_RUN_ADD_AT_REDUCE = 2, MSM_RUN_TWO_WAVES = 4 };

template <class F>
struct MsmEngine {
  using HF = typename HostFieldOf<F>::type;
  enum { SLOTS = 12, SLOT_PTS = 64 * 32 };  // three proofs in flight (groth16.hip PROOF_RING) x four G1 MSMs
  int nslots = SLOTS;  // slots the device buffers are sized for (the G2 engine serves one MSM per proof: 3)
  XYZZ<F>* buckets = nullptr;  // SLOTS x cap_buckets: one bucket array per MSM in flight
  XYZZ<F>* segsum = nullptr;  // SLOTS x seg_cap: reductions of different slots may run on different streams
  XYZZ<F>* segw = nullptr;
  uint64_t seg_cap = 0;
  XYZZ<F>* heavy_partial = nullptr;  // SLOTS x (MSM_HPOOL + MSM_HNC_POOL) partial sums of heavy buckets (msm_impl.hpp)
  uint32_t* heavy_ticket = nullptr;  // SLOTS x MSM_HEAVY_CAP: workgroups of a split heavy bucket that have finished (k_accum_heavy)
  uint32_t* heavy_plan = nullptr;    // SLOTS x MSM_HPLAN_WORDS: k_heavy_plan's work list for k_accum_heavy_nc (msm_impl.hpp)
  // per-(window, job) sums converted to the host representation; several MSMs can be
  // in flight on the stream, each with its own slot, pinned host copy and event
  XYZZ<HF>* partial = nullptr;    // device, SLOTS x SLOT_PTS
  XYZZ<F>* tree_stage = nullptr;  // device, nslots x MSM_STAGE_PTS: slices of k_treesum's job lists (small plans)
  XYZZ<HF>* h_partial = nullptr;  // pinned host, SLOTS x SLOT_PTS
  hipEvent_t done[SLOTS] = {};      // partials landed in h_partial
  hipEvent_t acc_done[SLOTS] = {};  // bucket accumulation finished
  hipEvent_t pre[SLOTS] = {};         // recorded in front of the accumulation (the sort is complete)
  hipEvent_t heavy_done[SLOTS] = {};  // heavy-bucket kernels finished (side stream)
  hipEvent_t redo_done[SLOTS] = {};   // k_accum_redo finished (it still reads the sort)
  uint32_t* redo = nullptr;           // [0] = count, [1..] = buckets a call-free accumulation kernel left to k_accum_redo
  MsmPlan slot_plan[SLOTS];
  // how finish_host* waits for a slot's event: spinning inside hipEventSynchronize (one MSM or one proof by itself: latency)
  // or polling with sleeps (the batch prover sets false: its driving thread must not hold one of the rank's few CPUs)
  bool host_spin = true;
  uint64_t cap_buckets = 0;
  uint64_t min_buckets = 0;  // floor set by reserve_buckets
  ~MsmEngine() {
    release();
    destroy_events();
  }
  void release();         // frees the buffers (a re-allocation may follow); events stay valid
  // The redo lists and the heavy-bucket tickets are zeroed once at allocation and every complete MSM leaves zeros behind
  // (k_accum_redo / k_accum_heavy clear them).  An MSM that was abandoned half-way -- a prover error path, after
  // ctx->drain() -- may not have: re-zero them so that the slot's next MSM does not append to a stale list.
  hipError_t reset_transients();
  void destroy_events();
  bool has_shared = false;
  hipError_t reserve(uint64_t n, bool shared_too = false);
  hipError_t reserve_buckets(uint64_t buckets);  // at least this many buckets per slot
  // The buffers must serve at least n slots from now on.  A context starts with TWO (what an MSM by itself, the pipelined big
  // MSM and the exchange collectives use: slots 0 and 1) -- a rank of a split 2^26-term MSM then holds 3 GB of bucket arrays
  // instead of the 18 GB of the prover's twelve; a key setup asks for SLOTS.  Growing frees the buffers (the next reserve()
  // rebuilds them); call it between synchronous entry points only.
  hipError_t ensure_slots(int n);
  // device part: bucket accumulation + reduction down to per-window partials,
  // async copy of the partials to the host and an event; does not block
  // accumulation runs on `st`; the (low-occupancy, latency-bound) reduction runs on
  // `st_reduce` behind an event so it overlaps the next MSM's accumulation; the heavy-bucket kernels
  // (which own buckets the accumulation skips) run on `st_heavy` beside the accumulation when given:
  // with uniform scalars their list is empty, and on `st` the empty launches waited ~0.8 ms each for
  // a free workgroup slot before the next accumulation could start
  // bucket_slot >= 0 and != slot: add INTO the bucket array of that slot (filled by an earlier run with MSM_RUN_NO_REDUCE)
  // and reduce the sum of both MSMs; see run_device_multi in msm_impl.hpp
  hipError_t run_device(const MsmSort& sort, const Affine<F>* d_bases, hipStream_t st, hipStream_t st_reduce,
                        PhaseTimer* prof, int ph_accum, int ph_reduce, int slot = 0, hipStream_t st_heavy = nullptr,
                        int bucket_slot = -1, int flags = 0, int bucket_slot3 = -1);
  // nm <= 4 MSMs of one plan shape (sorts[m] may repeat: several tables over one sort), each with its own slot and
  // reduction stream; the G1 call-free kernels accumulate them in one launch (one small proof: A, B1, L and H side by side)
  hipError_t run_device_multi(const MsmSort* const* sorts, const Affine<F>* const* d_bases, int nm, hipStream_t st,
                              const hipStream_t* st_reduces, PhaseTimer* prof, int ph_accum, int ph_reduce, const int* slots,
                              hipStream_t st_heavy = nullptr, const int* bucket_slots = nullptr, int flags = 0, int third_slot = -1);
  // host part: wait for the slot's event and combine (O(255) doublings on the CPU)
  hipError_t finish_host(XYZZ<HF>* out, int slot = 0);
  // per-window sums only (multi-GPU split: SURVEY.md §8e), nwin XYZZ points
  hipError_t finish_host_windows(XYZZ<HF>* out_windows, int slot = 0);
  static void windows_from_partials(const MsmPlan& pl, const XYZZ<HF>* h, XYZZ<HF>* out_windows, bool parallel = false);
  static int partials_per_msm(const MsmPlan& pl);  // XYZZ<HF> points a reduction writes per MSM (device `partial`, slot-major)
  // batched shared sort (MsmSort::run_shared_batch): one result per scalar vector
  hipError_t finish_host_batch(XYZZ<HF>* out, int slot = 0);
  // the same in two steps, so that the combination of the vectors' partition sums can run on several threads: wait for
  // the slot's partials, then the result of vector v (reads the pinned slot only)
  hipError_t wait_slot(int slot);
  XYZZ<HF> host_result_vec(int slot, int v) const;
};

// host-format affine points (Montgomery, R = 2^384) -> device format (R = 2^392 limbs)
template <class F>
hipError_t bases_convert(const Affine<typename HostFieldOf<F>::type>* d_in, Affine<F>* d_out, uint64_t n, hipStream_t st);

template <class HF>
XYZZ<HF> msm_combine_windows(const XYZZ<HF>* windows, int nwin, int c);

// table[w * n + i] = 2^(c w) * bases[i], w < ndigits (device format, affine)
template <class F>
hipError_t msm_build_table(const Affine<F>* d_bases, uint64_t n, const MsmPlan& plan, Affine<F>** out_table,
                           hipStream_t st);

extern template struct MsmEngine<Fq28>;
extern template struct MsmEngine<Fq2_28>;
extern template struct MsmEngine<BnFq28>;
extern template hipError_t msm_build_table<Fq28>(const Affine<Fq28>*, uint64_t, const MsmPlan&, Affine<Fq28>**, hipStream_t);
extern template hipError_t msm_build_table<Fq2_28>(const Affine<Fq2_28>*, uint64_t, const MsmPlan&, Affine<Fq2_28>**, hipStream_t);
extern template hipError_t msm_build_table<BnFq28>(const Affine<BnFq28>*, uint64_t, const MsmPlan&, Affine<BnFq28>**, hipStream_t);
extern template hipError_t bases_convert<Fq28>(const Affine<Fq>*, Affine<Fq28>*, uint64_t, hipStream_t);
extern template hipError_t bases_convert<Fq2_28>(const Affine<Fq2>*, Affine<Fq2_28>*, uint64_t, hipStream_t);
extern template hipError_t bases_convert<BnFq28>(const Affine<BnFq>*, Affine<BnFq28>*, uint64_t, hipStream_t);

}  // namespace zkmi

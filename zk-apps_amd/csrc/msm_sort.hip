// zkmi — bucket scatter for the Pippenger MSM: signed-digit decomposition of
// the scalars and a counting sort of point indices by (window, bucket).
// See msm_impl.hpp for the full kernel chain and HBM layout.
#include "msm_impl.hpp"

namespace zkmi {

namespace {

__device__ __forceinline__ void load_scalar(const uint32_t* __restrict__ scalars, uint32_t i, uint32_t* k) {
  const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  uint4 a = q[0], b = q[1];
  k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w;
  k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
}

}  // namespace

__global__ void __launch_bounds__(256)
k_hist(const uint32_t* __restrict__ scalars, uint32_t n, int c, int nwin, uint32_t nb,
       uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  load_scalar(scalars, i, k);
  DigitIter it;
  it.init(k, c);
  for (int w = 0; w < nwin; w++) {
    bool neg;
    uint32_t d = it.next(w, neg);
    if (d) atomicAdd(&counts[(uint32_t)w * nb + (d - 1)], 1u);
  }
}

// single-workgroup exclusive scan of counts[0..total) (in place); counts[total]
// receives the grand total; cursor = copy of the offsets.
__global__ void __launch_bounds__(1024)
k_scan(uint32_t* __restrict__ counts, uint32_t* __restrict__ cursor, uint32_t total) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (total + 1023u) / 1024u;
  const uint32_t beg = tid * per;
  const uint32_t end = (beg + per < total) ? beg + per : total;
  uint32_t s = 0;
  for (uint32_t i = beg; i < end; i++) s += counts[i];
  part[tid] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over 1024 partials
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = (tid == 0) ? 0u : part[tid - 1];
  for (uint32_t i = beg; i < end; i++) {
    uint32_t v = counts[i];
    counts[i] = run;
    cursor[i] = run;
    run += v;
  }
  if (tid == 1023) counts[total] = part[1023];
}

__global__ void __launch_bounds__(256)
k_scatter(const uint32_t* __restrict__ scalars, uint32_t n, int c, int nwin, uint32_t nb,
          uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  load_scalar(scalars, i, k);
  DigitIter it;
  it.init(k, c);
  for (int w = 0; w < nwin; w++) {
    bool neg;
    uint32_t d = it.next(w, neg);
    if (d) {
      uint32_t pos = atomicAdd(&cursor[(uint32_t)w * nb + (d - 1)], 1u);
      sorted[pos] = i | (neg ? 0x80000000u : 0u);
    }
  }
}

// ---------------------------------------------------------------------------
static int pick_window(uint64_t n) {
  // signed digits: 2^(c-1) buckets per window; mean bucket load n / 2^(c-1)
  // kept >= ~16 so a thread-per-bucket accumulation has work per lane.
  if (n <= (1u << 8)) return 5;
  if (n <= (1u << 11)) return 8;
  if (n <= (1u << 14)) return 10;
  if (n <= (1u << 17)) return 13;
  if (n <= (1u << 22)) return 16;
  return 17;
}

MsmPlan msm_make_plan_c(uint64_t n, int c) {
  MsmPlan p;
  p.c = c;
  p.nwin = 255 / p.c + 1;
  p.nb = 1u << (p.c - 1);
  p.n = n;
  return p;
}
MsmPlan msm_make_plan(uint64_t n) { return msm_make_plan_c(n, pick_window(n)); }

static const uint64_t PLAN_STEPS[] = {1u << 8, 1u << 11, 1u << 14, 1u << 17, 1u << 22, ~0ull};

uint64_t msm_max_buckets(uint64_t n) {
  uint64_t best = 0;
  for (uint64_t step : PLAN_STEPS) {
    uint64_t m = step < n ? step : n;
    MsmPlan p = msm_make_plan(m);
    uint64_t v = (uint64_t)p.nwin * p.nb;
    if (v > best) best = v;
    if (step >= n) break;
  }
  return best;
}

static uint64_t msm_max_entries(uint64_t n) {
  uint64_t best = 0;
  for (uint64_t step : PLAN_STEPS) {
    uint64_t m = step < n ? step : n;
    MsmPlan p = msm_make_plan(m);
    uint64_t v = (uint64_t)p.nwin * m;
    if (v > best) best = v;
    if (step >= n) break;
  }
  return best;
}

void MsmSort::release() {
  if (counts) (void)hipFree(counts);
  if (cursor) (void)hipFree(cursor);
  if (sorted) (void)hipFree(sorted);
  counts = cursor = sorted = nullptr;
  cap_entries = cap_buckets = 0;
}

hipError_t MsmSort::reserve(uint64_t n) {
  const uint64_t ne = msm_max_entries(n), nbk = msm_max_buckets(n);
  if (ne <= cap_entries && nbk <= cap_buckets) return hipSuccess;
  release();
  hipError_t e;
  if ((e = hipMalloc(&counts, sizeof(uint32_t) * (nbk + 1))) != hipSuccess) return e;
  if ((e = hipMalloc(&cursor, sizeof(uint32_t) * nbk)) != hipSuccess) return e;
  if ((e = hipMalloc(&sorted, sizeof(uint32_t) * (ne ? ne : 1))) != hipSuccess) return e;
  cap_entries = ne;
  cap_buckets = nbk;
  return hipSuccess;
}

hipError_t MsmSort::run(const uint32_t* d_scalars, uint64_t n, hipStream_t st, PhaseTimer* prof) {
  plan = plan_override ? msm_make_plan_c(n, plan_override) : msm_make_plan(n);
  const uint32_t tot_b = plan.nwin * plan.nb;
  hipError_t e;
  if ((e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * (tot_b + 1), st)) != hipSuccess) return e;
  const int T = 256;
  if (prof) prof->begin(PH_MSM_SORT, st);
  if (n) {
    hipLaunchKernelGGL(k_hist, dim3((n + T - 1) / T), dim3(T), 0, st, d_scalars, (uint32_t)n, plan.c, plan.nwin,
                       plan.nb, counts);
  }
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, counts, cursor, tot_b);
  if (n) {
    hipLaunchKernelGGL(k_scatter, dim3((n + T - 1) / T), dim3(T), 0, st, d_scalars, (uint32_t)n, plan.c, plan.nwin,
                       plan.nb, cursor, sorted);
  }
  if (prof) prof->end(PH_MSM_SORT, st);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
PhaseTimer::~PhaseTimer() {
  for (int i = 0; i < n_created; i++) {
    (void)hipEventDestroy(ev0[i]);
    (void)hipEventDestroy(ev1[i]);
  }
}
void PhaseTimer::begin(int phase, hipStream_t st) {
  if (!enabled || n_pending >= MAX_PENDING) return;
  if (n_pending >= n_created) {
    (void)hipEventCreate(&ev0[n_created]);
    (void)hipEventCreate(&ev1[n_created]);
    n_created++;
  }
  phase_of[n_pending] = phase;
  (void)hipEventRecord(ev0[n_pending], st);
}
void PhaseTimer::end(int phase, hipStream_t st) {
  if (!enabled || n_pending >= MAX_PENDING) return;
  (void)phase;
  (void)hipEventRecord(ev1[n_pending], st);
  n_pending++;
}
void PhaseTimer::collect() {
  for (int i = 0; i < n_pending; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev0[i], ev1[i]) == hipSuccess) {
      total_ms[phase_of[i]] += ms;
      count[phase_of[i]]++;
    }
  }
  n_pending = 0;
}
void PhaseTimer::reset() {
  n_pending = 0;
  for (int i = 0; i < 16; i++) {
    total_ms[i] = 0;
    count[i] = 0;
  }
}

}  // namespace zkmi

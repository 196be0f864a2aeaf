// zkmi — one big windowed G1 MSM (BASELINE config 3: 2^24 terms and more, 20-bit windows) as a two-stage pipeline.
//
// Sort, accumulation and reduction of such an MSM used to run back to back (27.6 + 122.5 + 5.7 ms at 2^26 terms): the digit
// sort is LDS- and memory-bound, the accumulation VALU-bound, and nothing overlapped.  Streams alone cannot overlap them --
// three 168-register accumulation waves hold 504 of a SIMD's 512 registers, so the sort's workgroups are not placed until the
// accumulation grid has drained -- and giving each its own CUs gains nothing (both are CU-time bound).  What works is
// co-residency on the SAME SIMDs:
//   group A = the upper windows (the partial top window among them), sorted first on the whole chip (2^15-bucket tiles);
//   its accumulation runs CAPPED at two waves per SIMD (k_accum_g1_nc_w2: 352 registers, same addition rate), and beside it,
//   on the front stream, the sort of group B = the lower windows with 2^14-bucket tiles (64 KB of LDS counters instead of 128)
//   -- 1 024-thread workgroups of <= 40 registers that fit the 160 registers and ~100 KB the capped accumulation leaves;
//   group B's accumulation then runs as usual.  Each group has its own digit-sort buffers (ctx->sort / ctx->sort_h), bucket
//   slot (0 / 1) and reduction stream.
#pragma once
#include "ctx.hpp"
#include "tune.hpp"

namespace zkmi {

// MEASURED NEUTRAL, so the product keeps the serial chain (A/B library: ZKMI_MSM_PIPE=1 selects the pipeline).  At 2^26 terms
// (profiles/r05/experiments/msm26_pipeline_ab.txt): the overlap happens -- group B's sort runs entirely beside group A's
// accumulation -- but the two-wave cap costs the accumulation 12.3 % by itself (136.9 against 121.9 ms for 13 windows:
// with a gather per iteration a third wave hides what two cannot), the co-resident sort another 10 % of group A's, and the
// sort itself takes 65 ms there instead of 13: 159.2 ms pipelined against 157.1 serial.
inline bool msm_pipe_applies(const MsmPlan& pl, uint64_t n) {
  return ZK_TUNE("ZKMI_MSM_PIPE", 0) != 0 && !pl.shared && pl.c > 16 && pl.nwin >= 8 && pl.nwin_total == 0 && n >= (1ull << 20);
}
// windows [0, wb) = group B, [wb, nwin) = group A
inline int msm_pipe_split(const MsmPlan& pl) { return pl.nwin / 2; }

// Queues both groups (nothing blocks).  Afterwards slot 0 holds group A's partial sums (windows [wb, nwin), reduction on
// stream_aux), slot 1 group B's (windows [0, wb), reduction on stream_aux2).  The caller has reserved ctx->sort, ctx->sort_h
// and ctx->g1 for the plan's size.
inline hipError_t msm_pipe_enqueue(zkmi_ctx* ctx, const uint32_t* d_scalars, uint64_t n, const Affine<Fq28>* bases, const MsmPlan& pl) {
  const int nwin = pl.nwin, wb = msm_pipe_split(pl);
  hipError_t e;
  // group A: sort on the main stream, whole chip
  ctx->sort.plan_override = pl.c;
  ctx->sort.win_first = wb;
  ctx->sort.win_count = nwin - wb;
  e = ctx->sort.run(d_scalars, n, ctx->stream, ctx->timer());
  ctx->sort.plan_override = 0;
  ctx->sort.win_first = ctx->sort.win_count = 0;
  if (e != hipSuccess) return e;
  if ((e = hipEventRecord(ctx->ev_sort[0], ctx->stream)) != hipSuccess) return e;
  // group B: sort on the front stream behind A's sort, beside A's accumulation
  if ((e = hipStreamWaitEvent(ctx->stream_front, ctx->ev_sort[0], 0)) != hipSuccess) return e;
  ctx->sort_h.plan_override = pl.c;
  ctx->sort_h.win_first = 0;
  ctx->sort_h.win_count = wb;
  ctx->sort_h.big_nb_log = 14;
  e = ctx->sort_h.run(d_scalars, n, ctx->stream_front, ctx->timer());
  ctx->sort_h.plan_override = 0;
  ctx->sort_h.win_first = ctx->sort_h.win_count = 0;
  ctx->sort_h.big_nb_log = 15;
  if (e != hipSuccess) return e;
  if ((e = hipEventRecord(ctx->ev_sorth[0], ctx->stream_front)) != hipSuccess) return e;
  // accumulation A, capped at two waves per SIMD
  e = ctx->g1.run_device(ctx->sort, bases, ctx->stream, ctx->stream_aux, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, 0, nullptr, -1,
                         MSM_RUN_TWO_WAVES);
  if (e != hipSuccess) return e;
  // accumulation B behind its sort
  if ((e = hipStreamWaitEvent(ctx->stream, ctx->ev_sorth[0], 0)) != hipSuccess) return e;
  return ctx->g1.run_device(ctx->sort_h, bases, ctx->stream, ctx->stream_aux2, ctx->timer(), PH_MSM_ACCUM_G1, PH_MSM_REDUCE_G1, 1);
}

// host: the per-window sums of both groups in window order (nwin points)
inline hipError_t msm_pipe_finish_windows(zkmi_ctx* ctx, const MsmPlan& pl, G1XYZZ* out_windows) {
  const int wb = msm_pipe_split(pl);
  hipError_t e = ctx->g1.finish_host_windows(out_windows + wb, 0);
  if (e != hipSuccess) return e;
  return ctx->g1.finish_host_windows(out_windows, 1);
}

}  // namespace zkmi

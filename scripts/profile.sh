#!/bin/bash
# Regenerates every file under profiles/<round>/ in one command, on the GPU box, from the repo root:
#
#     bash scripts/profile.sh r06            # -> gpurun_out/prof_r06/{trace,pmc_*}/..., summaries in profiles/r06/
# Run it LAST in a round, after the final commit that touches zk-apps_amd/csrc: the PMC summary is stamped with the digest
# of those sources and bench.py compares it with the sources it runs (roofline.traffic_source.same_sources_as_this_run).
#
# rocprofv3 is always given the program itself after `--` (python3 bench.py ...), tracing and counter
# collection are separate runs, and the counters are split over passes that fit the hardware slots
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: /opt/skills/guides/MI355X_MICROARCH.md, PMC slots).
set -u
ROUND=${1:-r06}
OUT=gpurun_out/prof_$ROUND
DST=profiles/$ROUND
STEPS_TRACE=${STEPS_TRACE:-48}
STEPS_PMC=${STEPS_PMC:-3}
mkdir -p "$OUT" "$DST"
export TMPDIR=/tmp
BENCH="python3 bench.py --warmup 1 --no-cpu-baseline --no-secondary --pmc-summary none"

# 1. kernel trace + per-kernel statistics (durations of kernels on different streams overlap)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps "$STEPS_TRACE" > "$OUT/trace.log" 2>&1
# 2. counters, one pass each
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps "$STEPS_PMC" > "$OUT/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps "$STEPS_PMC" > "$OUT/pmc_write.log" 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- $BENCH --steps "$STEPS_PMC" > "$OUT/pmc_sq.log" 2>&1
timeout 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_icache" -- $BENCH --steps "$STEPS_PMC" \
  > "$OUT/pmc_icache.log" 2>&1

# 2b. every kernel alone on the chip (counter collection serialises dispatches): clock, waves per SIMD, issue rate
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d "$OUT/alone" -- $BENCH --steps "$STEPS_PMC" \
  > "$OUT/alone.log" 2>&1
python3 scripts/kernel_alone_report.py "$OUT/alone" > "$DST/kernels_alone_bench_steps${STEPS_PMC}.txt" 2>> "$OUT/alone.log"
# 2c. not under the profiler: the complete default bench line and the proof rate per domain size
python3 scripts/domain_sweep.py 12 22 > "$DST/domain_sweep.txt" 2> "$OUT/domain_sweep.err"

# 3. summaries that are small enough to commit
STATS=$(find "$OUT/trace" -name '*_kernel_stats.csv' | head -1)
TRACE=$(find "$OUT/trace" -name '*_kernel_trace.csv' | head -1)
[ -n "$STATS" ] && cp "$STATS" "$DST/kernel_stats_bench_steps${STEPS_TRACE}.csv"
[ -n "$TRACE" ] && python3 scripts/trace_timeline.py "$TRACE" "$DST/timeline_bench_steps${STEPS_TRACE}.txt"
python3 scripts/pmc_summary.py "$DST/pmc_summary_bench_steps${STEPS_PMC}.json" $((STEPS_PMC + 1)) "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_icache"
grep -h '^{' "$OUT/trace.log" | tail -1 > "$DST/bench_line_under_trace.json"
# the complete line again, now with traffic / instruction counts from the fresh PMC summary
python3 bench.py --pmc-summary "$DST/pmc_summary_bench_steps${STEPS_PMC}.json" > "$DST/bench_line_full.json" 2> "$OUT/bench_full.err"
# 4. BASELINE configs[3] behind the bench contract (with its own PMC passes: HBM traffic and instruction count of the 2^26-term
#    accumulation launch), and the torch-free twin of the headline run under the native HIP runtime
MSM="python3 bench.py --workload msm26 --steps 1 --warmup 1 --msm-pmc-summary none"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/msm_pmc_fetch" -- $MSM > "$OUT/msm_pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/msm_pmc_write" -- $MSM > "$OUT/msm_pmc_write.log" 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv \
  -d "$OUT/msm_pmc_sq" -- $MSM > "$OUT/msm_pmc_sq.log" 2>&1
python3 scripts/pmc_summary.py "$DST/pmc_summary_msm26_steps1.json" 2 "$OUT/msm_pmc_fetch" "$OUT/msm_pmc_write" "$OUT/msm_pmc_sq"
python3 bench.py --workload msm26 --msm-pmc-summary "$DST/pmc_summary_msm26_steps1.json" > "$DST/bench_line_msm26_n1.json" 2> "$OUT/bench_msm26.err"
gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/bench_prove.c -Lzk-apps_amd -lzkmi -L/opt/rocm/lib -lamdhip64 \
  -Wl,-rpath,$PWD/zk-apps_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/bench_prove && \
  ZKMI_BACKTRACE=1 timeout 900 /tmp/bench_prove --log-n 20 --proofs 20 --warmup 2 --churn ${CHURN_OPS:-2000} > "$DST/c_bench_native_runtime.log" 2>&1
# 5. the next-row kernels on the same library: BN254 MSM / NTT / KZG commitment, opening, grand product;
#    Poseidon-5; SHA-256; assignment generation (DESIGN.md 4.5-4.8 quote this file)
mkdir -p "$DST/experiments"
for S in "bn254_timing.py 20" poseidon_timing.py sha_timing.py witness_timing.py; do
  echo "== python3 scripts/$S"; timeout 300 python3 scripts/$S 2>&1 | grep -v amdgpu.ids; echo
done > "$DST/experiments/next_rows_timing.txt"
# gpurun only brings gpurun_out/ back: leave a copy of the summaries there
mkdir -p "gpurun_out/profiles_$ROUND" && cp -r "$DST/." "gpurun_out/profiles_$ROUND/"
ls -la "$DST"

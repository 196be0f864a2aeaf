#!/bin/bash
# A/B of workgroup size and striped grids for the accumulation kernels (ZKMI_ACCUM_BLOCK, ZKMI_ACCUM_ROUNDS):
# isolated MSMs under rocprofv3 (kernel durations + resident-wave counters in the same run) and the proof rate.
OUT=${1:-gpurun_out/occ_ab}
mkdir -p "$OUT"
export TMPDIR=/tmp
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so  # the switches exist in the A/B library only (csrc/tune.hpp)
for CFG in "256 0" "64 0" "256 4" "64 4" "64 2" "64 8"; do
  set -- $CFG
  TAG=b$1_r$2
  ZKMI_ACCUM_BLOCK=$1 ZKMI_ACCUM_ROUNDS=$2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv \
    -d "$OUT/alone_$TAG" -- python3 scripts/msm_alone.py > "$OUT/alone_$TAG.log" 2>&1
  ZKMI_ACCUM_BLOCK=$1 ZKMI_ACCUM_ROUNDS=$2 python3 bench.py --steps 10 --no-cpu-baseline --no-secondary --pmc-summary none \
    > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
  echo "$TAG: $(grep -o '"value": [0-9.]*' "$OUT/bench_$TAG.json" | head -1) proofs/s"
done
python3 scripts/occupancy_report.py "$OUT"

#!/bin/bash
# A/B of the bucket-accumulation kernel variants on one box: ZKMI_ACCUM = 0 (first generation, out-of-line
# doubling path), 2 / 3 (call-free kernels at 2 / 3 waves per SIMD), each with the operand-scanning and the
# product-scanning (libzkmi_fips.so) Montgomery products.  For every variant: the isolated MSM timings
# (scripts/quick_timing.py) and the proof rate (bench.py, 10 steps).  Usage: bash scripts/accum_ab.sh OUTDIR
OUT=${1:-gpurun_out/accum_ab}
mkdir -p "$OUT"
for LIB in libzkmi libzkmi_fips; do
  [ -f zk-apps_amd/$LIB.so ] || continue
  for MODE in 0 2 3; do
    TAG=${LIB}_accum$MODE
    ZKMI_LIB=$PWD/zk-apps_amd/$LIB.so ZKMI_ACCUM=$MODE python3 scripts/quick_timing.py 20 > "$OUT/quick_$TAG.log" 2>&1
    ZKMI_LIB=$PWD/zk-apps_amd/$LIB.so ZKMI_ACCUM=$MODE python3 bench.py --steps 10 --no-cpu-baseline --no-secondary --pmc-summary none \
      > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
    echo "$TAG: $(grep -o '"value": [0-9.]*' "$OUT/bench_$TAG.json" | head -1) proofs/s; $(grep 'msm_g1 2' "$OUT/quick_$TAG.log" | tail -1)"
    grep 'msm_g2 2' "$OUT/quick_$TAG.log" | tail -1
  done
done

#!/bin/bash
# A/B of the bucket-accumulation kernel variants on one box.  ZKMI_ACCUM (G1) / ZKMI_ACCUM_G2: 0 = first-generation
# kernels (out-of-line doubling path, 248-256 VGPRs), 2 / 3 = call-free kernels compiled for 2 / 3 waves per SIMD;
# libzkmi_fips.so = the same sources with the product-scanning Montgomery products (make -C zk-apps_amd/csrc fips).
# For every variant: isolated MSM timings (scripts/quick_timing.py) and the proof rate (bench.py).
# Usage: bash scripts/accum_ab.sh OUTDIR ["lib g1 g2" ...]
OUT=${1:-gpurun_out/accum_ab}
shift
mkdir -p "$OUT"
CFGS=("$@")
# (the switches exist in the A/B library only: csrc/tune.hpp; libzkmi = the product, which ignores them)
[ ${#CFGS[@]} -eq 0 ] && CFGS=("libzkmi 3 2" "libzkmi_exp 0 0" "libzkmi_exp 2 2" "libzkmi_exp 3 2" "libzkmi_exp 3 3")
for CFG in "${CFGS[@]}"; do
  set -- $CFG
  LIB=$1; G1=$2; G2=$3
  [ -f zk-apps_amd/$LIB.so ] || continue
  TAG=${LIB}_g1m${G1}_g2m${G2}
  export ZKMI_LIB=$PWD/zk-apps_amd/$LIB.so ZKMI_ACCUM=$G1 ZKMI_ACCUM_G2=$G2
  python3 scripts/quick_timing.py 20 > "$OUT/quick_$TAG.log" 2>&1
  python3 bench.py --steps 12 --no-cpu-baseline --no-secondary --pmc-summary none > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
  echo "$TAG: $(grep -o '"value": [0-9.]*' "$OUT/bench_$TAG.json" | head -1) proofs/s"
  grep 'msm_g1 2' "$OUT/quick_$TAG.log" | tail -1
  grep 'msm_g2 2' "$OUT/quick_$TAG.log" | tail -1
done

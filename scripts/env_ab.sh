#!/bin/bash
# Interleaved A/B of environment-selected variants of the prover on ONE box: every variant is a string of
# VAR=value assignments; ROUNDS passes over all variants so that clock / thermal drift hits them alike.
# The switches only exist in the A/B library (csrc/tune.hpp, `make -C zk-apps_amd/csrc experiments`): every variant runs
# zk-apps_amd/libzkmi_exp.so (ZKMI_AB_LIB overrides), except the variant "product", which runs the product library.
# Usage: bash scripts/env_ab.sh OUTDIR ROUNDS "VAR=a VAR2=b" "VAR=c" ...     ("-" = no variables: the A/B library's defaults)
# Prints proofs/s (bench.py --steps 16, no secondaries) per variant and pass, then the per-variant median.
OUT=${1:-gpurun_out/env_ab}
ROUNDS=${2:-3}
shift 2
mkdir -p "$OUT"
VARIANTS=("$@")
STEPS=${STEPS:-16}
for R in $(seq 1 "$ROUNDS"); do
  I=0
  for V in "${VARIANTS[@]}"; do
    I=$((I + 1))
    ASSIGN="ZKMI_LIB=${ZKMI_AB_LIB:-$PWD/zk-apps_amd/libzkmi_exp.so}"
    [ "$V" != "-" ] && ASSIGN="$ASSIGN $V"
    [ "$V" = "product" ] && ASSIGN="ZKMI_LIB=$PWD/zk-apps_amd/libzkmi.so"
    env $ASSIGN python3 bench.py --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-secondary --pmc-summary none \
      > "$OUT/v${I}_r${R}.json" 2> "$OUT/v${I}_r${R}.err"
    VAL=$(grep -o '"value": [0-9.]*' "$OUT/v${I}_r${R}.json" | head -1 | cut -d' ' -f2)
    echo "pass $R variant $I [$V]: ${VAL:-FAILED} proofs/s"
  done
done
python3 - "$OUT" "$ROUNDS" "${VARIANTS[@]}" <<'EOF'
import json, statistics, sys
out, rounds, variants = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
for i, v in enumerate(variants, 1):
    vals = []
    for r in range(1, rounds + 1):
        try:
            vals.append(json.loads(open(f"{out}/v{i}_r{r}.json").read().strip().splitlines()[-1])["value"])
        except Exception:
            pass
    if vals:
        print(f"median [{v}]: {statistics.median(vals):.2f} proofs/s  (n={len(vals)}, min {min(vals):.2f}, max {max(vals):.2f})")
    else:
        print(f"median [{v}]: no result")
EOF

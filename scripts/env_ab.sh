#!/bin/bash
# Interleaved A/B of environment-selected variants of the prover on ONE box: every variant is a string of
# VAR=value assignments; ROUNDS passes over all variants so that clock / thermal drift hits them alike.
# Usage: bash scripts/env_ab.sh OUTDIR ROUNDS "VAR=a VAR2=b" "VAR=c" ...     ("-" = no variables: the defaults)
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
    ASSIGN=""
    [ "$V" != "-" ] && ASSIGN="$V"
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

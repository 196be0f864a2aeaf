#!/bin/bash
# scripts/ubench_affine.hip twice: plainly (time per addition) and under rocprofv3 --pmc (SQ_INSTS_VALU per dispatch ->
# wave-instructions per addition).  Usage: bash scripts/ubench_affine.sh OUTDIR [log2 pairs = 22]
OUT=${1:-gpurun_out/ubench_affine}
LG=${2:-22}
mkdir -p "$OUT" scripts/_bin
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-apps_amd/csrc scripts/ubench_affine.hip -o scripts/_bin/ubench_affine || exit 1
scripts/_bin/ubench_affine "$LG" 5 | tee "$OUT/plain.txt"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc" -- scripts/_bin/ubench_affine "$LG" 1 > "$OUT/pmc.txt" 2>&1
python3 - "$OUT" "$LG" <<'PY' | tee "$OUT/instructions.txt"
import csv, glob, sys
out, lg = sys.argv[1], int(sys.argv[2])
pairs = 1 << lg
rows = {}
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_pair" not in k:
            continue
        rows.setdefault((k, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for (k, d), c in sorted(rows.items(), key=lambda t: int(t[0][1])):
    seen.setdefault(k, c)  # first dispatch of each kernel
print(f"wave-instructions per addition (SQ_INSTS_VALU / (pairs / 64)), pairs = 2^{lg}:")
for k, c in seen.items():
    name = k.split("(")[0].replace("void ", "")
    v = c.get("SQ_INSTS_VALU", 0) / (pairs / 64)
    s = c.get("SQ_INSTS_SALU", 0) / (pairs / 64)
    print(f"  {name:34s} VALU {v:8.1f}  SALU {s:7.1f}")
PY

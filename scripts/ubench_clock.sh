#!/bin/bash
# scripts/ubench_clock.hip twice: plainly (clock, power, rates: 1.5 s per body) and under rocprofv3 --pmc (SQ_INSTS_VALU and
# busy cycles per dispatch: 0.2 s per body), then the table DESIGN.md section 3 quotes.  Usage: bash scripts/ubench_clock.sh OUTDIR [W]
OUT=${1:-gpurun_out/ubench_clock}
W=${2:-3}
mkdir -p "$OUT" scripts/_bin
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zk-apps_amd/csrc scripts/ubench_clock.hip -o scripts/_bin/ubench_clock || exit 1
scripts/_bin/ubench_clock 1.5 "$W" | tee "$OUT/plain_w$W.txt"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_w$W" -- scripts/_bin/ubench_clock 0.2 "$W" > "$OUT/pmc_w$W.txt" 2>&1
python3 scripts/ubench_clock_report.py "$OUT" "$W" | tee "$OUT/table_w$W.txt"

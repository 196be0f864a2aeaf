cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/g14
cat > /tmp/g14.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import bench, torch
z = bench.load_pkg().Zkmi(); ctx = z.context(0)
r = bench.small_domain_rate(z, ctx, "poseidon", 14, 1024)
print({k: r[k] for k in ("proofs_per_s", "single_proof_latency_ms", "host_cpus_busy")})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g14 -- python3 /tmp/g14.py 2>&1 | grep proofs_per_s
S=$(find gpurun_out/g14 -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%5.1f%%  calls %6s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
python3 /tmp/g14.py 2>&1 | grep proofs_per_s

python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
z = bench.load_pkg().Zkmi(); ctx = z.context(0)
for rel in ("chain", "poseidon", "chain", "poseidon"):
    for lg in (13, 14, 15):
        r = bench.small_domain_rate(z, ctx, rel, lg, 1024 if lg < 15 else 512)
        print(rel, lg, round(r['proofs_per_s'],1), round(r['single_proof_latency_ms'],2), flush=True)
PY

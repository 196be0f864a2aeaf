cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_sizes.py -x -q -m gpu -k "shared_gpu_dry_run or oracle_side_witness" 2>&1 | tail -25
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2; do for M in 3 6; do
echo "NTT_RB=$M"; ZKMI_NTT_RB=$M python - <<'PY'
import os, sys, time, statistics
sys.path.insert(0, ".")
import bench, torch
z = bench.load_pkg().Zkmi(os.environ.get("ZKMI_LIB")); ctx = z.context(0)
for lg in (16, 20, 22):
    r = bench.ntt_alone(z, ctx, lg, "none")
    print(lg, {k: round(v["ms"], 4) for k, v in r.items() if isinstance(v, dict) and "ms" in v})
PY
ZKMI_NTT_RB=$M timeout 600 python scripts/quad_ab.py single14 single20 batch20 2>&1 | grep "^{"
done; done

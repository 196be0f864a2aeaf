set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q1
ZKMI_DEBUG=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_split or degenerate or two_queries or witness_like or all_scalars_equal or golden or structured or bn254_msm" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_sizes.py -x -q -m gpu -k "msm_g1_2p20 or proof_bytes and not 22" 2>&1 | tail -5
for Q in 15 0; do
  echo "== ZKMI_QUAD=$Q domain sweep"
  ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so ZKMI_QUAD=$Q timeout 600 python scripts/domain_sweep.py 12 16 2>&1 | tail -8
done
STEPS=24 bash scripts/env_ab.sh gpurun_out/q1/ab 3 "ZKMI_QUAD=0" "ZKMI_QUAD=15" "ZKMI_QUAD=14" "ZKMI_QUAD=12" "product" 2>&1 | tail -25

cd $GRAFT_REPO_ROOT
ZKMI_DEBUG=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_split or degenerate or two_queries or witness_like or all_scalars_equal or golden or structured or bn254_msm or groth16" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_sizes.py -x -q -m gpu -k "msm_g2_2p18 or at_the_relations_own_sizes or oracle_side_witness" 2>&1 | tail -5
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2; do
for V in "0 0" "15 0" "15 15" "15 12"; do
  set -- $V
  echo "G2=$1 G2_BATCH=$2"; ZKMI_QUAD_G2=$1 ZKMI_QUAD_G2_BATCH=$2 timeout 600 python scripts/quad_ab.py single14 single20 group14 batch20 2>&1 | grep "^{"
done
done

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "degenerate or golden or witness_like or groth16 or mock_flow or structured" 2>&1 | tail -4
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2 3; do
for V in "1 1" "0 0" "1 0" "0 1"; do
  set -- $V
  echo "SPLIT_G2=$1 DEFER=$2"
  for lg in 12 13 14 15 16; do
    ZKMI_SOLO_SPLIT_G2=$1 ZKMI_HEAVY_DEFER=$2 timeout 300 python scripts/single_proof_trace.py $lg 2>&1 | grep latencies | sed "s/^/  2^$lg /"
  done
done
done

cd $GRAFT_REPO_ROOT
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
ZKMI_QUAD=31 ZKMI_QUAD_G2=31 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "groth16 or mock_flow or update_note_poseidon or grouped" 2>&1 | tail -4
for R in 1 2; do
for V in "15 15" "31 15" "15 31" "31 31"; do
  set -- $V
  echo "QUAD=$1 G2=$2"
  for lg in 12 13 14 15 16; do
    ZKMI_QUAD=$1 ZKMI_QUAD_G2=$2 timeout 300 python scripts/single_proof_trace.py $lg 2>&1 | grep latencies | sed "s/^/  2^$lg /"
  done
done
done

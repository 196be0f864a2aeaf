cd $GRAFT_REPO_ROOT
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2; do
for S in 0 2 3 4; do
  echo "SEG_LOG=$S"; ZKMI_SEG_LOG=$S timeout 600 python scripts/quad_ab.py single14 msm 2>&1 | grep "^{"
done
done
for lg in 12 13 15 16; do
for S in 0 3; do
  echo "lg=$lg SEG_LOG=$S"; ZKMI_SEG_LOG=$S timeout 300 python scripts/single_proof_trace.py $lg 2>&1 | grep latencies
done
done

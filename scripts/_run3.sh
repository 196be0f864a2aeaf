cd $GRAFT_REPO_ROOT
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2; do
for V in "0 0" "15 12" "15 15" "15 0" "15 8"; do
  set -- $V
  ZKMI_QUAD=$1 ZKMI_QUAD_BATCH=$2 timeout 600 python scripts/quad_ab.py 2>&1 | grep "^{"
done
done
for V in "0 0" "15 12" "15 15"; do
  set -- $V
  echo "QUAD=$1 BATCH=$2"; ZKMI_QUAD=$1 ZKMI_QUAD_BATCH=$2 timeout 600 python scripts/bits_relation_ab.py 20 12 2>&1 | grep "bits relation"
done

cd $GRAFT_REPO_ROOT
STEPS=32 bash scripts/env_ab.sh gpurun_out/ab14 6 "ZKMI_QUAD_BATCH=12 ZKMI_QUAD_G2_BATCH=12" "ZKMI_QUAD_BATCH=14 ZKMI_QUAD_G2_BATCH=12" "ZKMI_QUAD_BATCH=14 ZKMI_QUAD_G2_BATCH=14" "ZKMI_QUAD_BATCH=12 ZKMI_QUAD_G2_BATCH=14" 2>&1 | tail -8
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2 3; do for V in "12 12" "14 12" "14 14"; do set -- $V
 ZKMI_QUAD_BATCH=$1 ZKMI_QUAD_G2_BATCH=$2 timeout 300 python scripts/quad_ab.py group14 2>&1 | grep "^{"
done; done

cd $GRAFT_REPO_ROOT
export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for R in 1 2 3; do for S in 0 5 6 7; do
 echo -n "SEG_LOG=$S "; ZKMI_SEG_LOG=$S timeout 300 python scripts/quad_ab.py group14 2>&1 | grep "^{"
done; done
for S in 0 5 6; do echo "SEG_LOG=$S"; ZKMI_SEG_LOG=$S timeout 600 python scripts/domain_sweep.py 12 16 2>&1 | grep -v amdgpu; done

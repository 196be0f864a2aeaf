export ZKMI_LIB=$PWD/zk-apps_amd/libzkmi_exp.so
for r in 1 2; do
for v in 0 1; do
echo "== LH_MERGE_GROUPS=$v pass $r"; ZKMI_LH_MERGE_GROUPS=$v python3 scripts/domain_sweep.py 13 17 2>/dev/null | tail -5
done; done

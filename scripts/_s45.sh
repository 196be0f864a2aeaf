set -u
O=gpurun_out/s45; mkdir -p $O
for v in 19 21 19 21; do
  echo "== ZKMI_WIN_TWO_LEVEL=$v" >> $O/bn.log
  ZKMI_LIB=zk-apps_amd/libzkmi_exp.so ZKMI_WIN_TWO_LEVEL=$v timeout 300 python3 scripts/bn254_timing.py 20 2>&1 | grep "msm_g1 2^20" | head -4 >> $O/bn.log
  ZKMI_LIB=zk-apps_amd/libzkmi_exp.so ZKMI_WIN_TWO_LEVEL=$v timeout 300 python3 scripts/msm_scaling.py 19 21 2>&1 | grep -v amdgpu | tail -3 >> $O/bn.log
done

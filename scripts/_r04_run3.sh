set -x
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "big_window or point_split or multi_ctx or c_bench or msm_g1_2p26" 2>&1 | tail -15 | tee gpurun_out/r04/t3.log
python bench.py --workload msm26 --steps 3 --warmup 1 > gpurun_out/r04/bench_msm26.json 2> gpurun_out/r04/bench_msm26.err; echo "msm26 rc=$?"
cat gpurun_out/r04/bench_msm26.json | cut -c1-1500
tail -3 gpurun_out/r04/bench_msm26.err
python scripts/msm_scaling.py 22 26 2>&1 | tail -20 | tee gpurun_out/r04/msm_scaling.txt

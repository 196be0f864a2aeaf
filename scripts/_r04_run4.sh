set -x
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or multigpu_script or big_window or point_split or c_bench or msm26" 2>&1 | tail -15 | tee gpurun_out/r04/t4.log
python bench.py --workload msm26 --steps 3 --warmup 1 > gpurun_out/r04/bench_msm26.json 2> gpurun_out/r04/bench_msm26.err; echo "msm26 rc=$?"
cat gpurun_out/r04/bench_msm26.json | cut -c1-400; python -c "
import json; o=json.load(open('gpurun_out/r04/bench_msm26.json')); print(o['ms_per_step'], o['phase_ms_per_msm'], o['matches_closed_form_on_every_rank'])"
tail -3 gpurun_out/r04/bench_msm26.err

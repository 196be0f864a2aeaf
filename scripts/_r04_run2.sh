set -x
mkdir -p gpurun_out/r04
gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/bench_prove.c -Lzk-apps_amd -lzkmi -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zk-apps_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/bench_prove
ZKMI_BACKTRACE=1 timeout 1500 /tmp/bench_prove --log-n 20 --proofs 20 --warmup 2 --churn 2000 > gpurun_out/r04/c_bench_native_runtime.log 2>&1; echo "c_bench rc=$?" >> gpurun_out/r04/c_bench_native_runtime.log
tail -5 gpurun_out/r04/c_bench_native_runtime.log
python bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench_line.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r04/bench_line.json
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sizes.py -m gpu -x -q -k "c_bench or ab_switches or grouped_small_domain or churn_short or prove_batch_multi" 2>&1 | tail -15 | tee gpurun_out/r04/t2.log

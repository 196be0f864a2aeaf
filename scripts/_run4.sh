cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "2d_split" 2>&1 | tail -5
mkdir -p gpurun_out/t20
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t20 -- python3 scripts/single_proof_trace.py 20 2>&1 | grep latencies
T=$(find gpurun_out/t20 -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py "$T" gpurun_out/t20/timeline.txt 19.5 0.05 19.5
grep -v "^  s[0-9]* *k_\(order\|bucket\|part\|window\|heavy_plan\|check\|from_can\|quot\|entries\)" gpurun_out/t20/timeline.txt | head -150

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/t14
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t14 -- python3 scripts/single_proof_trace.py 14 2>&1 | tail -3
T=$(find gpurun_out/t14 -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py "$T" gpurun_out/t14/timeline.txt 2.6 0.0 2.6
cat gpurun_out/t14/timeline.txt | head -150

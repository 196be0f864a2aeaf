cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/t14b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t14b -- python3 scripts/single_proof_trace.py 14 2>&1 | grep latencies
T=$(find gpurun_out/t14b -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_timeline.py "$T" gpurun_out/t14b/timeline.txt 2.3 0.0 2.3
sed -n '/^timeline of/,$p' gpurun_out/t14b/timeline.txt
ZKMI_DEBUG=2 python3 scripts/single_proof_trace.py 14 2>&1 | grep "zkmi:" | tail -6

set -u
export TMPDIR=/tmp
O=gpurun_out/s33; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr14 -- python3 scripts/domain_sweep.py 14 14 > $O/tr14.log 2>&1
S=$(find $O/tr14 -name '*_kernel_stats.csv' | head -1); [ -n "$S" ] && cp $S $O/kernel_stats_2p14.csv
T=$(find $O/tr14 -name '*_kernel_trace.csv' | head -1); [ -n "$T" ] && python3 scripts/trace_timeline.py $T $O/timeline_2p14.txt all 0.02 400 > /dev/null
rm -rf $O/tr14
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr13 -- python3 scripts/domain_sweep.py 13 13 > $O/tr13.log 2>&1
S=$(find $O/tr13 -name '*_kernel_stats.csv' | head -1); [ -n "$S" ] && cp $S $O/kernel_stats_2p13.csv
rm -rf $O/tr13
python3 scripts/domain_sweep.py 12 16 > $O/sweep.txt 2>&1

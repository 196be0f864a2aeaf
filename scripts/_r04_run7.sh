export TMPDIR=/tmp
mkdir -p gpurun_out/r04/small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/small/trace13 -- python3 scripts/domain_sweep.py 13 13 > gpurun_out/r04/small/t13.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/small/trace14 -- python3 scripts/domain_sweep.py 14 14 > gpurun_out/r04/small/t14.log 2>&1
for d in 13 14; do f=$(find gpurun_out/r04/small/trace$d -name '*_kernel_stats.csv' | head -1); echo "== 2^$d"; head -22 "$f" | cut -c1-150; done
tail -2 gpurun_out/r04/small/t13.log gpurun_out/r04/small/t14.log

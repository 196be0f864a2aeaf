export TMPDIR=/tmp
mkdir -p gpurun_out/r04/small2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/small2/trace14 -- python3 scripts/domain_sweep.py 14 14 > gpurun_out/r04/small2/t14.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/small2/trace13 -- python3 scripts/domain_sweep.py 13 13 > gpurun_out/r04/small2/t13.log 2>&1
grep -h "^ *1[34] " gpurun_out/r04/small2/t14.log gpurun_out/r04/small2/t13.log

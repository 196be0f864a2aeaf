#!/bin/bash
# usage: clkpoll.sh TAG ENVASSIGN...   runs bench with --steps 300 and polls rocm-smi sclk/power every 0.5 s
TAG=$1; shift
env "$@" python3 bench.py --steps 300 --warmup 2 --no-cpu-baseline --no-secondary --pmc-summary none > gpurun_out/r3_z/bench_$TAG.json 2> gpurun_out/r3_z/bench_$TAG.err &
PID=$!
sleep 6
for i in $(seq 1 8); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3 | tr '\n' ' '
  echo
  sleep 0.5
done > gpurun_out/r3_z/smi_$TAG.txt
wait $PID
grep -o '"value": [0-9.]*' gpurun_out/r3_z/bench_$TAG.json | head -1

python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r04_launcher_n1.json 2> gpurun_out/r04_launcher_n1.err; echo "rc=$?"
cut -c1-300 gpurun_out/r04_launcher_n1.json; tail -3 gpurun_out/r04_launcher_n1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --workload msm26 --msm-log-n 22 --steps 2 --warmup 1 > gpurun_out/r04_launcher_msm.json 2> gpurun_out/r04_launcher_msm.err; echo "rc=$?"
cut -c1-300 gpurun_out/r04_launcher_msm.json; tail -3 gpurun_out/r04_launcher_msm.err
python bench.py --workload msm26 --msm-log-n 24 --split windows --steps 2 --warmup 1 2>/dev/null | cut -c1-400

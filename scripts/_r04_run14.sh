timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grouped or batch or churn or groth16_golden or mock_flow or update_note or c_bench or unsatisfied or every_domain or msm or canonical or reports" 2>&1 | tail -6
python3 scripts/domain_sweep.py 12 20 2>/dev/null | tail -10
python3 bench.py --no-cpu-baseline --steps 20 2>/dev/null | python3 -c "import json,sys; o=json.loads(sys.stdin.readline()); print('bench', o['value'], o['ms_per_step'], o['single_proof_latency_ms'], o['small_domain']['proofs_per_s'], o['small_domain']['single_proof_latency_ms'])"

timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sizes.py -m gpu -x -q -k "degenerate or two_queries or grouped or big_window or witness_like or msm_g1_2p20 or heavy or sizes_sweep or golden or msm_g2" 2>&1 | tail -8
python3 scripts/domain_sweep.py 12 17 2>/dev/null | tail -7

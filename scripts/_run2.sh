cd $GRAFT_REPO_ROOT
ZKMI_DEBUG=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_split" 2>&1 | grep -v "^$" | grep "zkmi_selftest\|passed\|failed"

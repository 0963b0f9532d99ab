cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "reclaimed or resident or out_of_memory or several_threads" 2>&1 | tail -3

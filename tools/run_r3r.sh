timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -x -q -m gpu -k "gkr or sharded" --durations=5 2>&1 | tail -12

timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench.py -x -q -m gpu --durations=8 > gpurun_out/r3c_tests.txt 2>&1; tail -25 gpurun_out/r3c_tests.txt

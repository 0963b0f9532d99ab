cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4g_gpu_suite.log 2>&1; tail -14 gpurun_out/r4g_gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r4g_bench.err | tail -1 > gpurun_out/r4g_bench_line.json; cut -c1-200 gpurun_out/r4g_bench_line.json

timeout 1500 python -m pytest tests -q -m gpu -x --durations=10 > gpurun_out/r3k_gputest.txt 2>&1; tail -16 gpurun_out/r3k_gputest.txt
timeout 600 python bench.py 2>gpurun_out/r3k_bench.err | tail -1 > gpurun_out/r3k_bench_line.json; cut -c1-250 gpurun_out/r3k_bench_line.json; grep bench gpurun_out/r3k_bench.err | tail -3

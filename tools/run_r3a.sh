free -g | head -2 > gpurun_out/r3a_host.txt; nproc >> gpurun_out/r3a_host.txt; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" >> gpurun_out/r3a_host.txt; df -h /tmp | tail -1 >> gpurun_out/r3a_host.txt
cat gpurun_out/r3a_host.txt
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r3a_gputest.txt 2>&1; tail -25 gpurun_out/r3a_gputest.txt
timeout 600 python bench.py 2>gpurun_out/r3a_bench.err | tail -1 > gpurun_out/r3a_bench_line.json; cut -c1-400 gpurun_out/r3a_bench_line.json; tail -3 gpurun_out/r3a_bench.err

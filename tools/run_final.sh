# round-3 final measurement set on HEAD: GPU suite, smoke, bench line, rocprofv3 stats + HBM counters, round times, other configs
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3z_gputest.txt 2>&1; grep -E "passed|failed" gpurun_out/r3z_gputest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile.sh r3z 2>&1 | tail -2
python tools/collect_profiles.py r3z 2>&1 | tail -1
timeout 600 python bench.py 2>gpurun_out/r3z_bench.err | tail -1 > gpurun_out/r3z_bench_line.json; cut -c1-260 gpurun_out/r3z_bench_line.json
timeout 120 python tools/round_times.py 24 2>&1 | tail -27 > gpurun_out/r3z_round_times.txt
timeout 900 python tools/bench_configs.py --config4 > gpurun_out/r3z_bench_configs.json 2>/dev/null; grep -c gpu_ms gpurun_out/r3z_bench_configs.json
timeout 300 python tools/interactive_time.py 8 12 16 20 2>&1 | grep nv= > gpurun_out/r3z_interactive.txt
cp profiles/r3z_hbm_traffic.json profiles/hbm_traffic_latest.json gpurun_out/ 2>/dev/null
cp profiles/r3z_rocprofv3_kernel_stats.csv profiles/r3z_rocprofv3_kernel_trace.csv gpurun_out/ 2>/dev/null

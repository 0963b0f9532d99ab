cd $GRAFT_REPO_ROOT
bash tools/profile.sh r1m > gpurun_out/profile_r1m.log 2>&1; tail -2 gpurun_out/profile_r1m.log
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r1m.log; cut -c1-250 gpurun_out/bench_r1m.log
timeout 300 python tools/round_times.py 24 2>&1 | tail -26 > gpurun_out/round_times_r1m.txt; tail -2 gpurun_out/round_times_r1m.txt

timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r2_gputest.txt 2>&1; grep -E "passed|failed" gpurun_out/r2_gputest.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r2e_bench_line.json; cut -c1-200 gpurun_out/r2e_bench_line.json
bash tools/profile.sh r2e 2>&1 | tail -3
timeout 120 python tools/round_times.py 24 2>&1 | tail -27 > gpurun_out/r2e_round_times.txt
timeout 900 python tools/bench_configs.py --config4 > gpurun_out/r2e_bench_configs.json 2>/dev/null; grep -c gpu_ms gpurun_out/r2e_bench_configs.json
SC_GKR_TRACE=1 timeout 200 python tools/bench_configs.py --only-gkr 2>&1 | grep "^\[gkr\]" | tail -7 > gpurun_out/r2e_gkr_stage_trace.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2egkr -o gkr -- python $R/tools/bench_configs.py --only-gkr > $R/gpurun_out/prof_r2egkr.log 2>&1; cd $R; find gpurun_out/prof_r2egkr -name "*.db" -delete
timeout 200 python tools/gkr_init_times.py 2>&1 | grep "dim " > gpurun_out/r2e_gkr_init_times.txt
timeout 200 python tools/oneshot_time.py 2>&1 | grep "nv=" > gpurun_out/r2e_oneshot_times.txt
timeout 100 python tools/evaluate_time.py 2>&1 | grep evaluate > gpurun_out/r2e_evaluate_times.txt; timeout 100 python tools/evaluate_time.py 20 2>&1 | grep evaluate >> gpurun_out/r2e_evaluate_times.txt
timeout 100 python tools/fix_variables_time.py 2>&1 | grep fix_ > gpurun_out/r2e_fix_variables_times.txt; timeout 100 python tools/fix_variables_time.py 18 2>&1 | grep fix_ >> gpurun_out/r2e_fix_variables_times.txt

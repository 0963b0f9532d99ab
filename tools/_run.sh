cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['all_kernels_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_t5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_t5/m1 -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_t5/m1.log 2>&1
grep -E "k_finalize|k_sum_combos|k_fix_multi|k_round" $R/gpurun_out/prof_t5/m1/t_kernel_stats.csv | cut -c1-60,150-260

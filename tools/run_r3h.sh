timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r3h_gputest.txt 2>&1; tail -14 gpurun_out/r3h_gputest.txt
for rep in 1 2; do for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
done; done
timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,28p'
for sh in c3 gkr; do SC_SHAPE=$sh SC_LIB_PATH=$PWD/tools/ab/tail_clocks.so timeout 200 python tools/tail_clocks.py 12 2>&1 | grep -v amdgpu; done
timeout 300 python tools/interactive_time.py 8 12 16 2>&1 | grep nv=
timeout 300 python tools/bench_configs.py --only-gkr 2>/dev/null | grep -E "gpu_ms|config" | head

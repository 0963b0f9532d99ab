# round-3 measurement set (HEAD): bench line, rocprofv3 stats + HBM counters, round times, other configs, VALU instruction counts
timeout 600 python bench.py 2>gpurun_out/r3j_bench.err | tail -1 > gpurun_out/r3j_bench_line.json; cut -c1-300 gpurun_out/r3j_bench_line.json; tail -2 gpurun_out/r3j_bench.err
bash tools/profile.sh r3j 2>&1 | tail -3
timeout 120 python tools/round_times.py 24 2>&1 | tail -27 > gpurun_out/r3j_round_times.txt
timeout 900 python tools/bench_configs.py --config4 > gpurun_out/r3j_bench_configs.json 2>/dev/null; grep -c gpu_ms gpurun_out/r3j_bench_configs.json
SC_GKR_TRACE=1 timeout 200 python tools/bench_configs.py --only-gkr 2>&1 | grep "^\[gkr\]" | tail -7 > gpurun_out/r3j_gkr_stage_trace.txt
bash tools/sq_valu_ab.sh r3j prev=tools/ab/libsumcheck_hip_prev.so cur=sumcheck_amd/libsumcheck_hip.so
timeout 200 python tools/oneshot_time.py 2>&1 | grep "nv=" > gpurun_out/r3j_oneshot_times.txt
timeout 300 python tools/interactive_time.py 8 12 16 20 2>&1 | grep nv= > gpurun_out/r3j_interactive.txt
for sh in c3 gkr; do SC_SHAPE=$sh SC_LIB_PATH=$PWD/tools/ab/tail_clocks.so timeout 200 python tools/tail_clocks.py 12 2>&1 | grep -v amdgpu; done > gpurun_out/r3j_tail_clocks.txt
for v in "" "SC_F29=0"; do echo "exp build $v"; env SC_LIB_VARIANT=exp $v timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4))"; done > gpurun_out/r3j_f29_ab.txt 2>&1
cat gpurun_out/r3j_f29_ab.txt

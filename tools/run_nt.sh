# nontemporal F29 loads + stores as the production path: GPU suite, then same-box A/B against the previous build
timeout 2200 python -m pytest tests -q -m gpu > gpurun_out/nt_suite.txt 2>&1; grep -E "passed|failed" gpurun_out/nt_suite.txt
for rep in 1 2 3; do
  for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,12p'; SC_LIB_PATH=$PWD/$L timeout 600 python tools/bench_configs.py --config4 2>/dev/null | grep -E "gpu_ms_median" | tr '\n' ' '; echo; done

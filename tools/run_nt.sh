# streaming by round traffic (>= 256 MiB) vs streaming always (previous build): parity subset, config 3, the small configs
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "random_shapes or config2 or claim_identity or golden or gkr or full_size" 2>&1 | grep -E "passed|failed" | tail -1
for rep in 1 2 3; do
  for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
    SC_LIB_PATH=$PWD/$L timeout 600 python tools/bench_configs.py 2>/dev/null | grep -E "gpu_ms_median" | head -3 | tr '\n' ' '; echo
  done
done

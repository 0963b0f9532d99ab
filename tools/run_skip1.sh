# node 1 from the claim identity: parity suite on the new library, then same-box A/B against the previous build
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
  for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,12p'; done

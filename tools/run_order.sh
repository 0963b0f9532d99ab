# A/B: alternate rounds dispatch the product rows in opposite order (tools/ab/altrows.so) vs the product library
timeout 600 env SC_LIB_PATH=$PWD/tools/ab/altrows.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_shapes or golden or config2 or claim_identity" 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do
  for L in sumcheck_amd/libsumcheck_hip.so tools/ab/altrows.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
for L in sumcheck_amd/libsumcheck_hip.so tools/ab/altrows.so; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,12p'; done

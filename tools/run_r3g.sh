# A/B: previous (r3b winner = commit 2) vs current (tail LDS metadata + hoisted slot loads) vs single-chain mads
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "random_shapes or golden or full_size_fiat or field_arithmetic or two_adic or interactive or gkr" 2>&1 | tail -3
for L in tools/ab/chain.so; do echo "== parity with $L"; SC_LIB_PATH=$PWD/$L timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "random_shapes or golden or full_size_fiat or field_arithmetic or two_adic" 2>&1 | tail -2; done
LIBS="tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so tools/ab/chain.so tools/ab/chain_m4.so"
for rep in 1 2 3; do
  for L in $LIBS; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
for L in sumcheck_amd/libsumcheck_hip.so tools/ab/chain.so; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,12p'; done
for sh in c3 gkr; do SC_SHAPE=$sh SC_LIB_PATH=$PWD/tools/ab/tail_clocks.so timeout 200 python tools/tail_clocks.py 12 2>&1 | grep -v amdgpu; done
timeout 300 python tools/interactive_time.py 8 12 16 2>&1 | grep nv=
timeout 300 python tools/bench_configs.py --only-gkr 2>/dev/null | tail -5

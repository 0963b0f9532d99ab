# A/B: flat-mode threshold by combination count (64 pairs for one product of 2-3 multiplicands) vs 16 pairs for every shape
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or random_shapes or gkr or interactive or config2" 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do
  for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo "$L"; SC_LIB_PATH=$PWD/$L timeout 300 python tools/bench_configs.py 2>/dev/null | grep -E "gpu_ms_median" | head -3 | tr '\n' ' '; echo
  done
done
for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 200 python tools/interactive_time.py 8 12 16 2>&1 | grep nv=; SC_LIB_PATH=$PWD/$L timeout 200 python tools/small_proofs.py 2>&1 | tail -6; done

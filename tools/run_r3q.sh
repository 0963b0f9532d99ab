LIBS="sumcheck_amd/libsumcheck_hip.so tools/ab/tail4k.so tools/ab/tail16k.so"
SC_LIB_PATH=$PWD/tools/ab/tail16k.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_shapes or golden or gkr" 2>&1 | tail -2
for rep in 1 2 3; do for L in $LIBS; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4))"
done; done
for L in $LIBS; do echo "== small proofs $L"; SC_LIB_PATH=$PWD/$L timeout 300 python tools/small_proofs.py 2>/dev/null | tail -2; SC_SHAPE=gkr SC_LIB_PATH=$PWD/$L timeout 300 python tools/small_proofs.py 2>/dev/null | tail -2; done
for L in $LIBS; do echo "== gkr $L"; SC_LIB_PATH=$PWD/$L timeout 300 python tools/bench_configs.py --only-gkr 2>/dev/null | grep gpu_ms_median_device; done

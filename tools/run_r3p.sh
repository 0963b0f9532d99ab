SC_LIB_PATH=$PWD/tools/ab/r1pref.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "random_shapes or golden or full_size_fiat or fuzz" 2>&1 | tail -2
LIBS="sumcheck_amd/libsumcheck_hip.so tools/ab/r1pref.so tools/ab/grid768.so tools/ab/r1pref_grid768.so"
for rep in 1 2 3 4; do for L in $LIBS; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), round(d['roofline']['avg_launch_ms'],4))"
done; done
for L in $LIBS; do echo "== $L"; SC_LIB_PATH=$PWD/$L timeout 120 python tools/round_times.py 24 2>&1 | sed -n '3,10p'; done

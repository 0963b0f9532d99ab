python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gkr" 2>&1 | tail -3
python tools/bench_configs.py --only-gkr 2>/dev/null | grep -E "gpu_ms"
echo "== baseline round times"; python tools/round_times.py 24 2>&1 | tail -27
for L in 0 30000; do for G in 768 512; do echo "== exp lib EXTRA_LDS=$L GRID=$G"; SC_LIB_VARIANT=exp SC_EXTRA_LDS=$L SC_GRID=$G python tools/round_times.py 24 2>&1 | sed -n '2,9p'; done; done

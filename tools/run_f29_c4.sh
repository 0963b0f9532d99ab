# config 4 (one product of three, HBM-bound, few multiplications per byte): 36-byte F29 bound tables vs canonical 32-byte ones (experiments build, SC_F29=0)
for rep in 1 2; do
  for F in 1 0; do
    echo "SC_F29=$F"; SC_LIB_VARIANT=exp SC_F29=$F timeout 600 python tools/bench_configs.py --config4 2>/dev/null | grep -E "\"config|gpu_ms_median" | tr '\n' ' '; echo
  done
done

# Does the shader clock hold under back-to-back proofs?  Samples rocm-smi while bench.py runs a long timed region.
# usage (GPU box): bash tools/clock_probe.sh [lib.so ...]   -> gpurun_out/clock_probe.txt
mkdir -p gpurun_out; OUT=gpurun_out/clock_probe.txt; : > $OUT
echo "== idle" >> $OUT; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|mclk\|power\|fclk" >> $OUT
LIBS="${@:-sumcheck_amd/libsumcheck_hip.so}"
for L in $LIBS; do
  echo "== $L" >> $OUT
  ( while :; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket Power" | tr '\n' ' '; echo; sleep 0.05; done ) > gpurun_out/clock_probe_$$.log &
  S=$!
  SC_LIB_PATH=$PWD/$L timeout 300 python bench.py --steps 5000 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4))" >> $OUT
  kill $S; wait $S 2>/dev/null
  cat gpurun_out/clock_probe_$$.log >> $OUT; rm -f gpurun_out/clock_probe_$$.log
done

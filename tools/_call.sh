cd $GRAFT_REPO_ROOT
for N in 4 8; do
  ( time SC_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus $N --steps 5 --warmup 2 2>gpurun_out/r4u_n${N}_proc.err | tail -1 > gpurun_out/r4u_n${N}_proc.json ) 2>&1 | grep real
done
( time SC_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 8 --steps 5 --warmup 2 --launcher threads 2>gpurun_out/r4u_n8_threads.err | tail -1 > gpurun_out/r4u_n8_threads.json ) 2>&1 | grep real
( time SC_BENCH_ONE_GPU=1 timeout 900 python3 bench.py --gpus 8 --config 4 --steps 3 --warmup 1 --launcher threads 2>gpurun_out/r4u_n8_c4_threads.err | tail -1 > gpurun_out/r4u_n8_c4_threads.json ) 2>&1 | grep real
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4u_*.json")):
    try:
        d = json.load(open(f)); c = d["cpu_baseline"] or {}
        print(f, d["n_gpus"], round(d["ms_per_step"], 2), "%.3g" % (c.get("value") or 0), c.get("cores"), c.get("cpus_allowed"), d["parity"].get("rounds_equal"), d["parity"]["ok"], d["config"]["ranks_seen"], d["config"]["communicator"], d["config"]["launcher"][:30])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r4u_n8_c4_threads.err

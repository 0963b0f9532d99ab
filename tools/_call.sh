cd $GRAFT_REPO_ROOT
export GIT_HEAD=1e9ae78
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -3
SC_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 2>gpurun_out/r4v_driver_n2.err | tail -1 > gpurun_out/r4v_driver_n2_one_gpu.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4v_driver_n2_one_gpu.json")); c = d["cpu_baseline"]
print("N=2 one GPU:", d["ms_per_step"], c["value"], c["cores"], c["cpus_allowed"], c["sample"][:60], d["parity"].get("rounds_equal"), d["parity"]["ok"])
PY
bash tools/measure.sh r4v 2>&1 | tail -12

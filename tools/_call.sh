cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -15
SC_BENCH_FORCE_SHARDED=1 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r4t_forced.err | tail -1 > gpurun_out/r4t_forced_rccl_one_rank.json; tail -5 gpurun_out/r4t_forced.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4t_forced_rccl_one_rank.json")); print(d["ms_per_step"], d["config"]["communicator"], d["config"]["exchange"], d["parity"])
PY

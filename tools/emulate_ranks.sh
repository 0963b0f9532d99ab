#!/bin/bash
# tools/emulate_ranks.sh TAG: the driver's multi-GPU command lines on ONE GPU (SC_BENCH_ONE_GPU=1: every rank on GPU 0) -- functional
# evidence that every N > 1 line carries parity, the communicator and the scaling model's prediction; not performance.
TAG=${1:?tag}
cd "$(dirname "$0")/.."
export SC_BENCH_ONE_GPU=1
run() { name=$1; shift; timeout 900 python3 bench.py "$@" --steps 5 --warmup 2 --min-gpu-seconds 0 > gpurun_out/$name.json 2> gpurun_out/$name.err; echo "$name rc=$?"; }
run ${TAG}_n2_proc --gpus 2
run ${TAG}_n4_proc --gpus 4
run ${TAG}_n8_threads --gpus 8 --launcher threads
run ${TAG}_n8_c4_threads --gpus 8 --launcher threads --config 4
TAG=$TAG python3 - <<'PY'
import json,glob,os
print("# file  n_gpus  ms_per_step  predicted_ms_per_step(real N GPUs)  exchange_assumed_us  parity.ok  ranks_seen  communicator  publication  launcher")
for f in sorted(glob.glob("gpurun_out/%s_n*.json" % os.environ["TAG"])):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    c=d["config"]
    print(f.split("/")[-1], d["n_gpus"], round(d["ms_per_step"],2), c.get("predicted_ms_per_step"), c.get("exchange_assumed_us"), d["parity"]["ok"], c.get("ranks_seen"), c.get("communicator"), (c.get("exchange") or {}).get("publication"), "|", c.get("launcher"))
PY

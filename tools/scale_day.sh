#!/bin/bash
# tools/scale_day.sh TAG [NMAX] -- the first multi-GPU lease in ONE call: everything that has never run between two devices, and the
# scaling model's verdict next to what was measured.  What the per-round exchange replaces is the rayon reduce of
# prover.rs:139-148.
#
#   1. the N >= 2 tests the one-GPU suite skips (RCCL one process per GPU, RCCL and peer-to-peer one thread per GPU, sharded GKR)
#   2. bench.py --gpus 1/2/4/8, config 3 (strong scaling: the metric as worded) -- each line runs sc_comm_selftest and
#      sc_comm_exchange_bench on its communicator before the proofs and carries predicted_ms_per_step:
#        a. one process per GPU over RCCL, direct publication (the default where the probe agrees on every rank)
#        b. the same with the publish kernel behind the all-reduce (--policy rccl_direct=0)
#        c. thread ranks over the library's peer-to-peer communicator (no collective library)
#   3. the gather threshold at NMAX: --policy shard_gather_log2 = 10 / 12 / 15
#   4. config 4 (nv = 28) at 1/2/4/8 on the faster of a / c
#   5. a table: measured against predicted, the exchange each line measured, parity
#
# With ONE visible GPU the script runs in emulation (SC_BENCH_ONE_GPU=1: every rank on GPU 0, the host transport standing in for RCCL) --
# that validates the script and the lines' fields, not performance.  Output: gpurun_out/TAG_scale_*.{json,err,log}, TAG_scale_summary.txt.
TAG=${1:?tag}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
NMAX=${2:-$NGPU}
EMU=0
if [ "$NGPU" -lt 2 ]; then EMU=1; NMAX=${2:-8}; export SC_BENCH_ONE_GPU=1; fi
STEPS=${SCALE_STEPS:-$([ $EMU = 1 ] && echo 4 || echo 50)}
NS=""; for n in 1 2 4 8; do [ $n -le $NMAX ] && NS="$NS $n"; done
echo "[scale_day] $NGPU visible GPUs, N in {$NS }, emulation=$EMU, $STEPS steps per line"

run() { # name, bench.py arguments
  name=${TAG}_scale_$1; shift
  timeout 1200 python3 bench.py "$@" --steps $STEPS --warmup 3 --min-gpu-seconds 0 --no-end-to-end > gpurun_out/$name.json 2> gpurun_out/$name.err
  echo "$name rc=$?"
}

# 1. the skipped tests (on one GPU they skip again: the count is what is recorded)
timeout 1800 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -k "one_process_per_gpu or one_thread_per_gpu or peer_to_peer_one_thread or gkr" \
  > gpurun_out/${TAG}_scale_tests.log 2>&1
tail -1 gpurun_out/${TAG}_scale_tests.log

# 2. config 3 at every N, three exchanges (N = 1 once: it has none)
run c3_n1 --gpus 1
for n in $NS; do
  [ $n = 1 ] && continue
  run c3_n${n}_rccl_direct  --gpus $n --launcher processes
  run c3_n${n}_rccl_publish --gpus $n --launcher processes --policy rccl_direct=0 --no-cpu-baseline
  run c3_n${n}_p2p          --gpus $n --launcher threads --no-cpu-baseline
done

# 3. the gather threshold at NMAX (default 15)
if [ $NMAX -ge 2 ]; then
  for g in 10 12; do
    run c3_n${NMAX}_rccl_gather$g --gpus $NMAX --launcher processes --policy shard_gather_log2=$g --no-cpu-baseline
    run c3_n${NMAX}_p2p_gather$g  --gpus $NMAX --launcher threads   --policy shard_gather_log2=$g --no-cpu-baseline
  done
fi

# 4. config 4 (nv = 28; in emulation nv = 22: eight shards of one GPU's memory are not the point)
C4NV=$([ $EMU = 1 ] && echo "--nv 22" || echo "")
for n in $NS; do
  if [ $n = 1 ]; then run c4_n1 --gpus 1 --config 4 $C4NV --no-cpu-baseline; continue; fi
  run c4_n${n}_rccl --gpus $n --config 4 $C4NV --launcher processes --no-cpu-baseline
  run c4_n${n}_p2p  --gpus $n --config 4 $C4NV --launcher threads --no-cpu-baseline
done

# 5. measured against the model
TAG=$TAG EMU=$EMU python3 - <<'PY' | tee gpurun_out/${TAG}_scale_summary.txt
import glob, json, os
tag, emu = os.environ["TAG"], os.environ["EMU"] == "1"
print("# scale_day summary%s" % (" -- ONE-GPU EMULATION: functional evidence only, every rank shares GPU 0" if emu else ""))
print("# line | N | ms_per_step | predicted | measured/predicted | speed-up vs N=1 | exchange_us (assumed by the model) | communicator / publication | policy | parity.ok | ranks_seen")
rows, base = [], {}
for f in sorted(glob.glob("gpurun_out/%s_scale_c*.json" % tag)):
    name = os.path.basename(f)[len(tag) + 7:-5]
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        err = open(f[:-5] + ".err").read().strip().splitlines()[-1:] if os.path.exists(f[:-5] + ".err") else []
        print("%s | no line: %s %s" % (name, e, err))
        continue
    rows.append((name, d))
    if d["n_gpus"] == 1:
        base[name[:2]] = d["ms_per_step"]
for name, d in rows:
    c = d["config"]
    ex = c.get("exchange") or {}
    b = base.get(name[:2])
    print("%s | %d | %.3f | %s | %s | %s | %s (%s) | %s / %s | %s | %s | %s" % (
        name, d["n_gpus"], d["ms_per_step"], c.get("predicted_ms_per_step") and round(c["predicted_ms_per_step"], 3),
        c.get("prediction", {}).get("measured_over_predicted") and round(c["prediction"]["measured_over_predicted"], 2),
        ("%.2fx" % (b / d["ms_per_step"])) if b else "-", ex.get("exchange_us") and round(ex["exchange_us"], 1), c.get("exchange_assumed_us"),
        c.get("communicator"), (ex.get("publication") or "-").split(":")[0], ",".join(c.get("policy") or []) or "-", d["parity"]["ok"], c.get("ranks_seen")))
if not emu:
    print("# verdict of the model: a line within 1.25x of predicted_ms_per_step confirms DESIGN 5.4 for that N; exchange_us against exchange_assumed_us says which term was off")
PY

#!/bin/bash
# Same-box A/B of library builds (one gpurun call; boxes differ by +-2 %, the order within a box does not matter).
#
#   tools/ab.sh [-k "<pytest -k expression>"] [-r REPS] [-w what,what,...] [-e "ENV=1 ENV2=x"] LIB [LIB ...]
#
#   LIB    a library file: sumcheck_amd/libsumcheck_hip.so, tools/ab/<variant>.so (tools/build_variant.sh NAME -DFLAG ...), or the
#          literal `exp` for the experiments build (SC_LIB_VARIANT=exp; combine with -e "SC_F29=0" and the like)
#   -k     run this slice of the GPU parity suite on every LIB first (a variant that is not bit-exact is not a candidate)
#   -w     what to time, comma separated (default bench,rounds):
#            bench        python bench.py --no-cpu-baseline: ms per proof (mean, min), average big-round launch     [config 3, nv=24]
#            rounds       tools/round_times.py 24: device and wall time per round   (-R "25 c4": other arguments, e.g. config 4's shard)
#            small        tools/small_proofs.py: whole proofs at nv 8..16 (latency-bound rounds)                     [SC_SHAPE=c3|c2|gkr]
#            configs      tools/bench_configs.py: BASELINE configs 2, README shape, 5                                (+ config 4 with configs4)
#            gkr          tools/bench_configs.py --only-gkr
#            wide         tools/bench_configs.py --wide: products of 5..12 multiplicands at nv=20
#            interactive  tools/interactive_time.py 8 12 16 20
#            tailclocks   tools/tail_clocks.py 12 (needs a -DSC_TAIL_CLOCKS build)
#   -r     repetitions of the bench / small / configs legs (default 3), interleaved across the LIBs
# Older builds that lack newer entry points load with SC_AB_ALLOW_MISSING=1 (set here).
cd "$(dirname "$0")/.."
K=""; REPS=3; WHAT="bench,rounds"; EXTRA_ENV=""; RARGS="24"
while getopts "k:r:w:e:R:" o; do case $o in k) K=$OPTARG;; r) REPS=$OPTARG;; w) WHAT=$OPTARG;; e) EXTRA_ENV=$OPTARG;; R) RARGS=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { sed -n '2,22p' "$0"; exit 2; }
export SC_AB_ALLOW_MISSING=1
libenv() { if [ "$1" = exp ]; then echo "SC_LIB_VARIANT=exp $EXTRA_ENV"; else echo "SC_LIB_PATH=$PWD/$1 $EXTRA_ENV"; fi; }
has() { case ",$WHAT," in *",$1,"*) return 0;; esac; return 1; }
line='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), round(d["roofline"]["avg_launch_ms"],4))'
cat /sys/fs/cgroup/cpu.max 2>/dev/null | head -1
if [ -n "$K" ]; then
  for L in "$@"; do echo "== parity [$K] $L"; env $(libenv $L) timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "$K" 2>&1 | grep -E "passed|failed|error" | tail -2; done
fi
for rep in $(seq 1 $REPS); do
  for L in "$@"; do
    if has bench; then echo -n "bench $L  "; env $(libenv $L) timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$line"; fi
    if has small; then echo "small $L"; env $(libenv $L) timeout 300 python tools/small_proofs.py 2>&1 | grep "nv="; fi
    if has configs; then echo -n "configs $L  "; env $(libenv $L) timeout 600 python tools/bench_configs.py 2>/dev/null | grep -E "gpu_ms_median" | tr '\n' ' '; echo; fi
    if has configs4; then echo -n "configs4 $L  "; env $(libenv $L) timeout 900 python tools/bench_configs.py --config4 2>/dev/null | grep -E "gpu_ms_median" | tr '\n' ' '; echo; fi
    if has wide; then echo "wide $L"; env $(libenv $L) timeout 600 python tools/bench_configs.py --wide 2>/dev/null | python -c 'import sys,json; [print("  ",k,round(v["gpu_ms_median"],3),round(v["gpu_ms_min"],3)) for k,v in json.load(sys.stdin).items()]'; fi
    if has gkr; then echo -n "gkr $L  "; env $(libenv $L) timeout 300 python tools/bench_configs.py --only-gkr 2>/dev/null | grep gpu_ms_median | tr '\n' ' '; echo; fi
  done
done
for L in "$@"; do
  if has rounds; then echo "== rounds $L"; env $(libenv $L) timeout 300 python tools/round_times.py $RARGS 2>&1 | sed -n '3,32p'; fi
  if has interactive; then echo "== interactive $L"; env $(libenv $L) timeout 300 python tools/interactive_time.py 8 12 16 20 2>&1 | grep nv=; fi
  if has tailclocks; then for sh in c3 gkr; do echo "== tail clocks $sh $L"; env $(libenv $L) SC_SHAPE=$sh timeout 200 python tools/tail_clocks.py 12 2>&1 | grep -v amdgpu; done; fi
done

#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters for bench.py's workload.
#   tools/profile.sh TAG [extra bench.py arguments, e.g. --config 4 --nv 25]
# Outputs under gpurun_out/prof_<tag>/ ; python tools/collect_profiles.py TAG copies the summaries into profiles/.
set -u
TAG=${1:-r1}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --min-gpu-seconds 0 $*"  # (--no-end-to-end: the host-table proofs after the clock launch chunk-sized round-1 kernels that are not the launches the line describes)
echo "$CMD" > $OUT/command.txt
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
PMC_CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --min-gpu-seconds 0 $*"  # (counters are per launch: a few proofs are enough)
echo "$PMC_CMD" > $OUT/command_pmc.txt
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $PMC_CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT

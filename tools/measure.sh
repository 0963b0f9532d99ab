#!/bin/bash
# The end-of-round measurement set on HEAD, one gpurun call:  gpurun -- "GIT_HEAD=$(git rev-parse --short HEAD) bash tools/measure.sh TAG"   (writes
# gpurun_out/TAG_*; copy what is to be judged into profiles/: TAG_hbm_traffic_latest.json as profiles/hbm_traffic_latest.json, TAG_rocprof_kernel_latest.json as
# profiles/rocprof_kernel_latest.json, TAG_bench_line.json also as profiles/bench_line_latest.json -- tests/test_host.py checks that the three agree).  GPU suite, smoke, rocprofv3 stats + HBM counter passes (tools/profile.sh), bench line, round times, the
# other BASELINE configs, the interactive protocol.
TAG=${1:?tag}
cd "$(dirname "$0")/.."
timeout 1800 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/${TAG}_gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_gpu_suite.log | tail -1
cp gpurun_out/plan_coverage.json gpurun_out/${TAG}_plan_coverage.json 2>/dev/null  # (the plan -> oracle-comparing tests table of THIS full run: tests/conftest.py; later partial runs overwrite the plain name)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile.sh $TAG 2>&1 | tail -2
python tools/collect_profiles.py $TAG 2>&1 | tail -1
# (what collect_profiles wrote into profiles/ on this box travels back through gpurun_out/ -- these files only: older profiles/${TAG}_* must not overwrite this run's)
cp profiles/${TAG}_rocprofv3_kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_trace.csv profiles/${TAG}_hbm_traffic.json profiles/${TAG}_rocprof_kernel.json gpurun_out/ 2>/dev/null; cp profiles/hbm_traffic_latest.json gpurun_out/${TAG}_hbm_traffic_latest.json 2>/dev/null; cp profiles/rocprof_kernel_latest.json gpurun_out/${TAG}_rocprof_kernel_latest.json 2>/dev/null
timeout 600 python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json; cut -c1-260 gpurun_out/${TAG}_bench_line.json
timeout 120 python tools/round_times.py 24 2>&1 | tail -27 > gpurun_out/${TAG}_round_times.txt
timeout 900 python tools/bench_configs.py --config4 > gpurun_out/${TAG}_bench_configs.json 2>/dev/null; grep -c gpu_ms gpurun_out/${TAG}_bench_configs.json
SC_GKR_TRACE=1 timeout 200 python tools/bench_configs.py --only-gkr 2>&1 | grep "^\[gkr\]" | tail -7 > gpurun_out/${TAG}_gkr_stage_trace.txt
timeout 300 python tools/interactive_time.py 8 12 16 20 2>&1 | grep nv= > gpurun_out/${TAG}_interactive.txt
timeout 200 python tools/oneshot_time.py 2>&1 | grep "nv=" > gpurun_out/${TAG}_oneshot_times.txt
bash tools/emulate_ranks.sh $TAG > gpurun_out/${TAG}_multi_rank_launches_one_gpu.txt 2>&1; tail -5 gpurun_out/${TAG}_multi_rank_launches_one_gpu.txt | cut -c1-200

cd $GRAFT_REPO_ROOT
export GIT_HEAD=182d11a
bash tools/profile.sh r4s 2>&1 | tail -2
python tools/collect_profiles.py r4s 2>&1 | tail -1
timeout 600 python bench.py 2>gpurun_out/r4s_bench.err | tail -1 > gpurun_out/r4s_bench_line.json; cut -c1-300 gpurun_out/r4s_bench_line.json
cp profiles/r4s_* profiles/hbm_traffic_latest.json gpurun_out/

cd $GRAFT_REPO_ROOT
export GIT_HEAD=b59c294
bash tools/measure.sh r4z 2>&1 | tail -12
cp profiles/r4z_* profiles/hbm_traffic_latest.json gpurun_out/ 2>/dev/null

cd $GRAFT_REPO_ROOT
export GIT_HEAD=d26e81e
bash tools/measure.sh r4n 2>&1 | tail -20
bash tools/sq_breakdown.sh > /dev/null 2>&1; python tools/sq_breakdown_summary.py r4n 9 2>&1 | tail -10
cp profiles/r4n_* profiles/hbm_traffic_latest.json gpurun_out/ 2>/dev/null

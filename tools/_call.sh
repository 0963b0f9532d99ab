cd $GRAFT_REPO_ROOT
# robustness pass on the round's final library: soak, differential fuzz with fresh seeds, the GPU suite twice more
timeout 900 python tools/soak.py > gpurun_out/r4y_soak.txt 2>&1; tail -6 gpurun_out/r4y_soak.txt
timeout 900 python tools/fuzz.py 250 777 > gpurun_out/r4y_fuzz.txt 2>&1; tail -3 gpurun_out/r4y_fuzz.txt
timeout 900 python tools/fuzz_eval_gkr.py 200 160 9191 > gpurun_out/r4y_fuzz_eval_gkr.txt 2>&1; tail -3 gpurun_out/r4y_fuzz_eval_gkr.txt
for i in 1 2; do timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r4y_suite_$i.log 2>&1; grep -E "passed|failed" gpurun_out/r4y_suite_$i.log | tail -1; done

python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gkr" 2>&1 | tail -5
python tools/bench_configs.py > gpurun_out/r2_bench_configs.json 2> gpurun_out/r2_bench_configs.err; tail -3 gpurun_out/r2_bench_configs.err
SC_GKR_TRACE=1 python tools/bench_configs.py --only-gkr 2>&1 | tail -60 > gpurun_out/r2_gkr_trace.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2gkr -o gkr -- python $R/tools/bench_configs.py --only-gkr > $R/gpurun_out/prof_r2gkr.log 2>&1; cd $R; find gpurun_out/prof_r2gkr -name "*.db" -delete; ls gpurun_out/prof_r2gkr/*

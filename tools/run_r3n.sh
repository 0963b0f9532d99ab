timeout 900 python tools/streamed_big.py 30 1 2>gpurun_out/r3n_streamed_big.err | tail -1 > gpurun_out/r3n_streamed_nv30.json; cat gpurun_out/r3n_streamed_nv30.json; tail -2 gpurun_out/r3n_streamed_big.err
timeout 900 python tools/fuzz.py 200 777 2>&1 | tail -3 > gpurun_out/r3n_fuzz.txt; cat gpurun_out/r3n_fuzz.txt
timeout 600 python tools/fuzz_eval_gkr.py 0 120 4321 2>&1 | tail -2 >> gpurun_out/r3n_fuzz.txt; timeout 600 python tools/fuzz_eval_gkr.py 150 0 99 2>&1 | tail -2 >> gpurun_out/r3n_fuzz.txt; tail -4 gpurun_out/r3n_fuzz.txt
timeout 900 python tools/soak.py 2>&1 | tail -6 > gpurun_out/r3n_soak.txt; cat gpurun_out/r3n_soak.txt

cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py tests/test_gpu_variants.py -q -m gpu -x --durations=6 > gpurun_out/r4d_tests.txt 2>&1; tail -10 gpurun_out/r4d_tests.txt
bash tools/ab.sh -k "golden or random_shapes or gkr or interactive" -w bench,small,gkr -r 3 tools/ab/notile.so sumcheck_amd/libsumcheck_hip.so tools/ab/tile8k.so tools/ab/tile16k.so tools/ab/tile64k.so > gpurun_out/r4d_tile_ab.txt 2>&1
SC_SHAPE=gkr bash tools/ab.sh -w small -r 2 tools/ab/notile.so sumcheck_amd/libsumcheck_hip.so tools/ab/solo1.so tools/ab/solo4.so >> gpurun_out/r4d_tile_ab.txt 2>&1
bash tools/ab.sh -w small -r 2 sumcheck_amd/libsumcheck_hip.so tools/ab/solo1.so tools/ab/solo4.so >> gpurun_out/r4d_tile_ab.txt 2>&1
bash tools/ab.sh -w interactive -r 0 tools/ab/notile.so sumcheck_amd/libsumcheck_hip.so >> gpurun_out/r4d_tile_ab.txt 2>&1
bash tools/ab.sh -w tailclocks -r 0 tools/ab/tileclk.so > gpurun_out/r4d_tail_clocks.txt 2>&1
grep -v "^$" gpurun_out/r4d_tile_ab.txt | cut -c1-300 | tail -120; cat gpurun_out/r4d_tail_clocks.txt | tail -34

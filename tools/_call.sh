cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x --durations=3 > gpurun_out/r4k_tests.txt 2>&1; tail -5 gpurun_out/r4k_tests.txt
bash tools/ab.sh -k "golden or random_shapes or full_size_fiat" -w bench,small -r 4 tools/ab/permad.so sumcheck_amd/libsumcheck_hip.so tools/ab/chainall.so > gpurun_out/r4k_chain_ab.txt 2>&1
bash tools/ab.sh -w rounds -r 0 sumcheck_amd/libsumcheck_hip.so tools/ab/chainall.so >> gpurun_out/r4k_chain_ab.txt 2>&1
bash tools/ab.sh -w configs4 -r 2 sumcheck_amd/libsumcheck_hip.so tools/ab/chainall.so >> gpurun_out/r4k_chain_ab.txt 2>&1
grep -v "^$" gpurun_out/r4k_chain_ab.txt | cut -c1-420 | sed 's/"gpu_ms_median_host_inputs_incl_h2d[^,]*,//'

cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_variants.py tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -4
bash tools/ab.sh -w bench,small,gkr -r 3 tools/ab/finold.so sumcheck_amd/libsumcheck_hip.so 2>&1 | grep -v "^$" | cut -c1-200 | sed 's/"gpu_ms_median_host_inputs_incl_h2d.*//'
SC_SHAPE=gkr bash tools/ab.sh -w small,interactive -r 2 tools/ab/finold.so sumcheck_amd/libsumcheck_hip.so 2>&1 | grep -v "^$" | cut -c1-200

cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo; done

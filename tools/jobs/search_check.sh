set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_range_search.py tests/test_gpu_range_wide.py tests/test_gpu_full_size.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
R=14336 C=4096 timeout -k 10 200 python3 tools/quick_bench.py 2>&1 | grep "range search"

set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/search
timeout -k 10 600 python -u -m pytest tests/test_gpu_range_search.py tests/test_gpu_range_wide.py tests/test_gpu_range_acts.py tests/test_gpu_range_products.py tests/test_gpu_full_size.py tests/test_gpu_llm_stage.py -x -q --timeout 300 --timeout-method thread > gpurun_out/search/tests.log 2>&1 || { tail -40 gpurun_out/search/tests.log; exit 1; }
tail -2 gpurun_out/search/tests.log
R=14336 C=4096 timeout -k 10 200 python3 tools/quick_bench.py 2>&1 | grep -i "range search\|fake_quant fused" | tee gpurun_out/search/quick.log

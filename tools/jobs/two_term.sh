set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_gptq.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4
timeout -k 10 120 python3 tools/leaf_check.py
timeout -k 10 120 python3 tools/quick_linalg.py inv 14336
timeout -k 10 600 python -u -m pytest tests/test_gpu_headline_tokens.py tests/test_gpu_full_size.py tests/test_gpu_llm_stage.py -x -q --timeout 280 --timeout-method thread 2>&1 | tail -3

set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_gptq.py tests/test_zz_no_alloc.py tests/test_gpu_full_size.py tests/test_gpu_headline_tokens.py tests/test_gpu_llm_stage.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8
STEPS=8 bash tools/jobs/ab.sh

set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_gptq.py tests/test_gpu_headline_tokens.py tests/test_gpu_full_size.py tests/test_gpu_llm_stage.py tests/test_gpu_parallel.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
STEPS=8 bash tools/jobs/bench2.sh

set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_gptq.py -x -q --timeout 120 --timeout-method thread -k "hessian" 2>&1 | tail -5
timeout -k 10 600 python -u -m pytest tests/test_gpu_headline_tokens.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
bash tools/jobs/bench2.sh

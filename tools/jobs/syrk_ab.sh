set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/syrk_ab
timeout -k 10 600 python -u -m pytest tests/test_gpu_gptq.py tests/test_gpu_headline_tokens.py tests/test_gpu_full_size.py tests/test_gpu_llm_stage.py -x -q --timeout 300 --timeout-method thread > gpurun_out/syrk_ab/tests.log 2>&1 || { tail -30 gpurun_out/syrk_ab/tests.log; exit 1; }
tail -2 gpurun_out/syrk_ab/tests.log
STEPS=8 bash tools/jobs/bench2.sh
bash tools/jobs/syrk_split_traffic.sh | grep -A12 '"false>\|"true>' | grep 'hbm_bytes\|launches\|SyrkParams'

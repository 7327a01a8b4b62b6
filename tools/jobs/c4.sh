set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout -k 10 400 python -u -m pytest tests/test_gpu_linear_scorer.py tests/test_gpu_smooth_fused.py -x -q --timeout 120 --timeout-method thread > gpurun_out/c4/tests.log 2>&1 || { tail -60 gpurun_out/c4/tests.log; exit 1; }
tail -2 gpurun_out/c4/tests.log
timeout -k 10 300 python3 tools/quick_c4.py 2>&1 | tee gpurun_out/c4/quick.log

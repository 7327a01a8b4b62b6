set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/all
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/all/tests.log 2>&1 || { tail -30 gpurun_out/all/tests.log; exit 1; }
tail -3 gpurun_out/all/tests.log

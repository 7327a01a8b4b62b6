set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/calib
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "smooth or range or reorder or lowrank or stage or fake_quant" > gpurun_out/calib/tests.log 2>&1 || { tail -30 gpurun_out/calib/tests.log; exit 1; }
tail -2 gpurun_out/calib/tests.log
bash tools/jobs/bench_quick.sh

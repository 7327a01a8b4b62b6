set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fqc
timeout -k 10 600 python -u -m pytest tests/test_gpu_fake_quant.py tests/test_gpu_smooth_fused.py tests/test_gpu_range_wide.py tests/test_gpu_smooth_tensor.py -x -q --timeout 120 --timeout-method thread > gpurun_out/fqc/tests.log 2>&1 || { tail -40 gpurun_out/fqc/tests.log; exit 1; }
tail -2 gpurun_out/fqc/tests.log
bash tools/jobs/fqv.sh

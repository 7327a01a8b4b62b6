set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_gptq.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 120 python3 tools/leaf_check.py | tail -2
cp deepcompressor_amd/libdcq.so /tmp/new.so
for i in 1 2; do
for v in /tmp/new.so tools/bin/libdcq_base.so; do
cp $v deepcompressor_amd/libdcq.so
echo "== $v"
timeout -k 10 120 python3 tools/quick_sweep_full.py
timeout -k 10 120 python3 tools/quick_linalg.py inv 14336
done
done
cp /tmp/new.so deepcompressor_amd/libdcq.so

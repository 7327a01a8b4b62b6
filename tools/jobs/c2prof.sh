set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c2prof -o c2 --output-format csv -- python3 tools/quick_c2.py > gpurun_out/c2prof/run.log 2>&1 || { tail -20 gpurun_out/c2prof/run.log; exit 1; }
find gpurun_out/c2prof -name "*kernel_stats.csv" | head -1 | xargs head -12

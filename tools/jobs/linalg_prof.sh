set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/la
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/la/prof -o l -- python3 tools/quick_linalg.py inv 14336 > gpurun_out/la/inv.log 2> gpurun_out/la/prof.err
cat gpurun_out/la/inv.log
python3 tools/prof_db.py $(find gpurun_out/la/prof -name "*.db" | head -1) 8

# Per-kernel times of the leaf and the pair node inside one inverse factor of 2048 columns (kernel trace).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/leafprof
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o l -- python3 tools/quick_linalg.py inv 2048 > $O/run.log 2>&1
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 12 | grep "pair_factor\|panel_factor" | cut -c1-90
find $O -name "*.db" -delete

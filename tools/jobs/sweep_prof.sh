set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweepprof
rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 tools/quick_sweep_full.py
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o l -- python3 tools/quick_sweep_full.py > $O/run.log 2>&1
tail -1 $O/run.log
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 14 | cut -c1-150
find $O -name "*.db" -delete

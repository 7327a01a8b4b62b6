set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/lowrank
rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 tools/quick_lowrank.py 5
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o l -- python3 tools/quick_lowrank.py 5 > $O/run.log 2>&1
cat $O/run.log | tail -1
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 25 | cut -c1-140
find $O -name "*.db" -delete

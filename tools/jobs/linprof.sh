set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/linprof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o l -- python3 tools/quick_linalg.py ${MODE:-inv 14336} > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
tail -3 $O/run.log
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 25 > $O/kernel_stats.txt
cut -c1-160 $O/kernel_stats.txt | head -20
find $O -name "*.db" -delete

set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 40 > $O/bench_kernel_stats.txt
python3 tools/trace_timeline.py $(find $O/prof -name "*.db" | head -1) > $O/bench_timeline.txt || echo "timeline failed"
find $O -name "*.db" -delete
tail -c 300 $O/bench_prof.json

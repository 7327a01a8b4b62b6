set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/bp/prof -o b -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/bp/bench.json 2> gpurun_out/bp/prof.err || { tail -30 gpurun_out/bp/prof.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/bp/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
python3 tools/trace_timeline.py $(find gpurun_out/bp/prof -name "*.db" | head -1) > gpurun_out/bp/timeline.txt
python3 tools/prof_db.py $(find gpurun_out/bp/prof -name "*.db" | head -1) 70 > gpurun_out/bp/stats.txt; find gpurun_out/bp -name "*.db" -delete

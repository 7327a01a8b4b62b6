set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/trace
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/prof -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $O/bench.json 2> $O/prof.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = open("gpurun_out/trace/trace_small.csv", "w")
out.write("start,end,stream,name\n")
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0][-60:]
    out.write(f'{r["Start_Timestamp"]},{r["End_Timestamp"]},{r.get("Stream_Id", r.get("Queue_Id"))},{short}\n')
out.close()
PY
rm -rf $O/prof
ls -la $O

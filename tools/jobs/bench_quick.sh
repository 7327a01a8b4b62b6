set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bq
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/bq/bench.json 2> gpurun_out/bq/bench.err || { tail -30 gpurun_out/bq/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bq/bench.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print(json.dumps(d["roofline"]))
for r in d["rooflines"]:
    print(r["kernel"][:70], round(r.get("achieved", 0)), r.get("frac"), r.get("avg_launch_ms"))
for r in d["other_configs"]:
    print(r)
PY

set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bq
timeout -k 10 800 python3 bench.py > gpurun_out/bq/bench_full.json 2> gpurun_out/bq/bench_full.err || { tail -30 gpurun_out/bq/bench_full.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bq/bench_full.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print(json.dumps(d.get("cpu_baseline"))[:1000])
for r in d["rooflines"]:
    print(r["kernel"][:60], round(r.get("achieved", 0)), r.get("frac"), r.get("avg_launch_ms"), r.get("default_policy_ms"))
for r in d["other_configs"]:
    print(r)
PY

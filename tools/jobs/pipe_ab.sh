# the bench step with and without the next step's Hessians queued beside the current step's pipelines
set -e
cd $GRAFT_REPO_ROOT
run() {
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-8} $2 > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'],1), 'frac', round(d['roofline']['frac'],3), 'launches', d['roofline']['launches'])" "$1"
}
run warm ""
for i in 1 2; do
run serial ""
run pipelined --pipeline
done

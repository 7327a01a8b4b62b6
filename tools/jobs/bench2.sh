set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-3} > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), round(d['roofline']['alone']['frac'],3))"
done

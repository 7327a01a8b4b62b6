set -e
cd $GRAFT_REPO_ROOT
run() {
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-8} > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'],1), round(d['roofline']['frac'],3))" "$1"
}
run warm
for i in 1 2; do
run base
for v in $VARIANTS; do
  export $v
  run "$v"
  unset ${v%%=*}
done
done

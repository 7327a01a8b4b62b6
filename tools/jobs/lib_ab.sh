# A/B of two builds of the library inside the bench step: the tree's own and $ALT (a path to another libdcq.so)
set -e
cd $GRAFT_REPO_ROOT
cp deepcompressor_amd/libdcq.so /tmp/base.so
run() {
cp $2 deepcompressor_amd/libdcq.so
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-8} > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json,sys
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'],1), round(d['roofline']['frac'],3))" "$1"
}
run warm /tmp/base.so
for i in 1 2; do
run base /tmp/base.so
run alt $ALT
done
cp /tmp/base.so deepcompressor_amd/libdcq.so

# A/B of the SYRK round width for the Hessians that run beside other pipelines (same box, alternating order)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/width
for W in ${WIDTHS:-0 240 224 0 192 248 208}; do
DCQ_EXP_WIDTH=$W timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-6} > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('width $W ms_per_step', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), round(d['roofline']['alone']['frac'],3))" | tee -a gpurun_out/width/log.txt
done

# the sweep alone and the bench step with other sizes of the lazy group (blocks of 128 columns per far update)
set -e
cd $GRAFT_REPO_ROOT
for lazy in 4 6 8; do
sed -i "s/^R, Cn, lazy = 4096, 14336, [0-9]*/R, Cn, lazy = 4096, 14336, $lazy/" tools/quick_sweep_full.py
sed -i "s/^LAZY_BLOCKS = [0-9]*/LAZY_BLOCKS = $lazy/" deepcompressor_amd/quantizer/kernel/gptq.py
echo "== lazy $lazy"
timeout -k 10 120 python3 tools/quick_sweep_full.py
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-8} > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1))"
done
sed -i "s/^LAZY_BLOCKS = [0-9]*/LAZY_BLOCKS = 4/" deepcompressor_amd/quantizer/kernel/gptq.py
sed -i "s/^R, Cn, lazy = 4096, 14336, [0-9]*/R, Cn, lazy = 4096, 14336, 4/" tools/quick_sweep_full.py

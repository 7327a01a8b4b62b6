set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_gptq.py tests/test_zz_no_alloc.py tests/test_gpu_headline_tokens.py tests/test_gpu_full_size.py tests/test_gpu_llm_stage.py tests/test_gpu_llm_layer_stages.py tests/test_gpu_parallel.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 > /tmp/b.json 2>/tmp/b.err || { tail /tmp/b.err; exit 1; }
python3 -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1), round(d['roofline']['frac'],3), round(d['roofline']['alone']['frac'],3))"
done

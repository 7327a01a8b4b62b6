set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/planes
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/pmc_sq -o s -- python3 tools/quick_linalg.py big > $O/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 tools/quick_linalg.py big > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -o t -- python3 tools/quick_linalg.py big > $O/pmc_tcc.log 2>&1
python3 tools/kernel_pmc.py $O/pmc.json "gemm_planes_kernel" $(find $O/pmc_* -name "*.db") > /dev/null
find $O -name "*.db" -delete
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/planes/pmc.json'))['kernels']
for k,v in d.items():
    a=v['avg_per_launch']; print(repr(k), v['launches_per_pass'], {c:round(x/1e6,2) for c,x in a.items()}, v['avg_duration_us_under_pmc'])
PY

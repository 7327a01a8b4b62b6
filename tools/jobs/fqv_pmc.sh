set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fqv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/fqv/pmc_fetch -o f -- python3 tools/fq_variants_rate.py > gpurun_out/fqv/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/fqv/pmc_write -o w -- python3 tools/fq_variants_rate.py > gpurun_out/fqv/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/fqv/pmc_sq -o s -- python3 tools/fq_variants_rate.py > gpurun_out/fqv/pmc_sq.log 2>&1
python3 tools/kernel_pmc.py gpurun_out/fqv/pmc.json "progressive_int_kernel,fused_table_kernel,tensor_absmax_kernel" $(find gpurun_out/fqv/pmc_* -name "*.db") > /dev/null
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/fqv/pmc.json'))['kernels']
for k,v in d.items():
    a=v['avg_per_launch']; print(repr(k), v['launches_per_pass'], {c:round(x/1e6,2) for c,x in a.items()}, v['avg_duration_us_under_pmc'].get('SQ_INSTS_VALU'), v.get('hbm_bytes'))
PY

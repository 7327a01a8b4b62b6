# HBM-side traffic of the SYRK per Hessian width: the transposed-copy kernel (C = 14336) and the direct kernel (C = 4096)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/syrk_split
mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/w.log 2>&1
python3 tools/kernel_pmc.py $O/syrk_split_pmc.json "false>(dcq::SyrkParams,true>(dcq::SyrkParams,transpose_pad_kernel" $(find $O/f $O/w -name "*.db") | tail -60
find $O -name "*.db" -delete

set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2a/tests.log 2>&1 || { tail -20 gpurun_out/r2a/tests.log; exit 1; }
tail -3 gpurun_out/r2a/tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -c 600 gpurun_out/r2a/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a/prof -o b -- python3 bench.py --no-cpu-baseline > gpurun_out/r2a/bench_prof.json 2> gpurun_out/r2a/prof.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2a/pmc_fetch -o f -- python3 tools/stage_kernels.py > gpurun_out/r2a/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2a/pmc_write -o w -- python3 tools/stage_kernels.py > gpurun_out/r2a/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/r2a/pmc_sq -o s -- python3 tools/stage_kernels.py > gpurun_out/r2a/pmc_sq.log 2>&1
find gpurun_out/r2a -name "*.csv" | head -20

# Final measurements of the round: GPU tests, bench line, kernel trace of the same bench command, counter passes.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k
mkdir -p $O
trap 'rc=$?; find $O -name "*.db" -delete; if [ $rc -ne 0 ]; then echo "failed at line $LINENO"; for f in $O/*.err $O/*.log; do echo "== $f"; tail -5 $f; done; fi' EXIT
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -3 $O/tests.log
fi
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
python3 tools/prof_db.py $(find $O/prof -name "*.db" | head -1) 40 > $O/bench_kernel_stats.txt
python3 tools/trace_timeline.py $(find $O/prof -name "*.db" | head -1) > $O/bench_timeline.txt || echo "timeline failed"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 tools/stage_kernels.py > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 tools/stage_kernels.py > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq -o s -- python3 tools/stage_kernels.py > $O/pmc_sq.log 2>&1
python3 tools/kernel_pmc.py $O/stage_pmc.json "fused_int_kernel,range_search_kernel,fused_table_kernel,progressive_int_kernel,tensor_absmax_kernel" $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq -name "*.db") > /dev/null
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/syrk_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/syrk_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/syrk_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/syrk_write.log 2>&1
python3 tools/kernel_pmc.py $O/syrk_pmc.json "syrk_kernel" $(find $O/syrk_fetch $O/syrk_write -name "*.db") > /dev/null
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r3k/syrk_pmc.json"))
k = d["kernels"]["syrk_kernel"]
hessians = 8  # bench.py --steps 1 --warmup 0: the step's 4 and the 4 of the pass with nothing else on the chip
out = {"kernel": "syrk_kernel: all rounds (kernel launches) of one dcq_hessian_accumulate call together; C = 14336 once "
                 "and C = 4096 three times in the step and again in the pass alone, T = 262144",
       "method": d["method"], "kernel_launches_per_pass": k["launches_per_pass"], "hessians_per_pass": hessians,
       "avg_per_kernel_launch": k["avg_per_launch"],
       "avg_hbm_bytes_per_launch": k["hbm_bytes"] * k["launches_per_pass"] / hessians}
json.dump(out, open("gpurun_out/r3k/syrk_traffic.json", "w"), indent=1)
print(json.dumps(out)[:400])
PY
find $O -name "*.db" -delete
echo r3k done

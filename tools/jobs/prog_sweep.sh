set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ps
for np in 4 2 1; do for nt in 0 1; do
rm -rf gpurun_out/ps/prof
DCQ_PROG_NP=$np DCQ_PROG_NT=$nt timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ps/prof -o f -- python3 tools/fq_variants_rate.py ${CASES:-qoq_g128} > gpurun_out/ps/rate.log 2> gpurun_out/ps/prof.err
echo "np=$np nt=$nt: $(cat gpurun_out/ps/rate.log | tr '\n' ' ')"
python3 tools/prof_db.py $(find gpurun_out/ps/prof -name "*.db" | head -1) 4 | grep "dcq::"
done; done
rm -rf gpurun_out/ps/prof

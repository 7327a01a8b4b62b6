set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ps
for passes in 1 4 8 32; do for nt in 0 1; do
rm -rf gpurun_out/ps/prof
DCQ_TABLE_PASSES=$passes DCQ_TABLE_NT=$nt timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ps/prof -o f -- python3 tools/fq_variants_rate.py nvfp4_g16 > gpurun_out/ps/rate.log 2> gpurun_out/ps/prof.err
echo "passes=$passes nt=$nt: $(cat gpurun_out/ps/rate.log | tr '\n' ' ')"
python3 tools/prof_db.py $(find gpurun_out/ps/prof -name "*.db" | head -1) 4 | grep "dcq::"
done; done
rm -rf gpurun_out/ps/prof

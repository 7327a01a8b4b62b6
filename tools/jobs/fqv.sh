set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fqv
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/fqv/prof -o f -- python3 tools/fq_variants_rate.py > gpurun_out/fqv/rate.log 2> gpurun_out/fqv/prof.err
cat gpurun_out/fqv/rate.log
python3 tools/prof_db.py $(find gpurun_out/fqv/prof -name "*.db" | head -1) 6
find gpurun_out/fqv -name "*.db" -delete

set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fqv
timeout -k 10 200 python3 tools/fq_variants_rate.py > gpurun_out/fqv/rate_plain.log 2>&1
cat gpurun_out/fqv/rate_plain.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/fqv/prof -o f -- python3 tools/fq_variants_rate.py > gpurun_out/fqv/rate.log 2> gpurun_out/fqv/prof.err
cat gpurun_out/fqv/rate.log
DB=$(find gpurun_out/fqv/prof -name "*.db" | head -1)
python3 tools/prof_db.py $DB 8
python3 tools/prof_seq.py $DB fused_int_kernel 12
python3 tools/prof_seq.py $DB fused_table_kernel 12
python3 tools/prof_seq.py $DB progressive_int_kernel 12
find gpurun_out/fqv -name "*.db" -delete

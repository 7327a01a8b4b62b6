# rate of the fake-quant variants with the tree's library and with $ALT, alternating
set -e
cd $GRAFT_REPO_ROOT
cp deepcompressor_amd/libdcq.so /tmp/new.so
for i in 1 2; do
for v in /tmp/new.so $ALT; do
cp $v deepcompressor_amd/libdcq.so
echo "== $v"
timeout -k 10 200 python3 tools/fq_variants_rate.py ${CASES:-nvfp4_g16}
done
done
cp /tmp/new.so deepcompressor_amd/libdcq.so

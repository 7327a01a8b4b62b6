set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 tools/quick_linalg.py err
timeout -k 10 200 python3 tools/quick_linalg.py
echo "== wide for K >= 128"
DCQ_WIDE_MIN_K=128 timeout -k 10 200 python3 tools/quick_linalg.py 2>&1 | grep -E "K=128 |K=512 "

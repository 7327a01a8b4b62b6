set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 tools/quick_linalg.py err
timeout -k 10 200 python3 tools/quick_linalg.py
DCQ_SPLIT_MIN=36 timeout -k 10 200 python3 tools/quick_linalg.py 2>&1 | grep -E "M=2048|M=1024|n=4096|n=14336"

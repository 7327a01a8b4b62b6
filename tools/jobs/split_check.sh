set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 tools/quick_linalg.py err
timeout -k 10 200 python3 tools/quick_linalg.py

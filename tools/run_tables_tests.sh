set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "tables or sharded or synthetic" 2>&1 | tail -15

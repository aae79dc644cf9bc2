set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "split_hpoly" --timeout 120 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20

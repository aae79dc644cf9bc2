set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 2>&1 | tail -3 && bash tools/run_b24.sh "$@"

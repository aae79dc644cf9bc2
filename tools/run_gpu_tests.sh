set -o pipefail
cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=16
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -40

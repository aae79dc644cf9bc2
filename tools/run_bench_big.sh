set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
( time timeout -k 10 900 python bench.py --log-domain 24 --steps 3 --warmup 1 ) > gpurun_out/bench24.json 2> gpurun_out/bench24.err; echo "rc=$?"; tail -5 gpurun_out/bench24.err; cat gpurun_out/bench24.json

# the 2^26 whole-proof test alone (BASELINE.json configs[3]'s circuit on one GPU)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 -k "configs3" > gpurun_out/r2_huge.log 2>&1; rc=$?
tail -15 gpurun_out/r2_huge.log; tail -12 gpurun_out/fullsize_progress.log
exit $rc

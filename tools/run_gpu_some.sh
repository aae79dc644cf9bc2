# run selected GPU tests: bash tools/run_gpu_some.sh <pytest args>
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 1000 python -m pytest "$@" -m gpu -x -q --durations=8 > gpurun_out/some_tests.log 2>&1; rc=$?
tail -40 gpurun_out/some_tests.log
exit $rc

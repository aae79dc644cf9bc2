# round 2, first GPU call: whole GPU suite (2^26 test skipped here) + default bench with --check
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
UG_HUGE_LOG=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r2_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --check > gpurun_out/r2_bench24.json 2> gpurun_out/r2_bench24.err; rc=$?
tail -3 gpurun_out/r2_bench24.err; cat gpurun_out/r2_bench24.json
exit $rc

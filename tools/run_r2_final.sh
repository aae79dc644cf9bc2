# round-end rehearsal: what the driver runs -- the whole GPU suite (2^26 included), smoke(), the default bench
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r2_final_tests.log 2>&1; rc=$?
tail -14 gpurun_out/r2_final_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
( time python bench.py ) > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err; rc=$?
tail -4 gpurun_out/r2_final_bench.err; cut -c1-700 gpurun_out/r2_final_bench.json
exit $rc

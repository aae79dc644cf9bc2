# staged witness buffers: the concurrency test, the sharded phase tests (they stage the same way), then the 2^24 bench
# with one and with two host threads:  bash tools/run_r2_pipe.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_registry.py -m gpu -x -q --durations=5 > gpurun_out/pipe_tests.log 2>&1; rc=$?
tail -15 gpurun_out/pipe_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/run_b24.sh --check --host-threads 1 && cp gpurun_out/q24.json gpurun_out/q24_threads1.json &&
bash tools/run_b24.sh --check --host-threads 2 && cp gpurun_out/q24.json gpurun_out/q24_threads2.json

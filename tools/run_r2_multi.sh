set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=8
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "slices or sharded or rccl" > gpurun_out/r2_sharded_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r2_sharded_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/run_multi.sh

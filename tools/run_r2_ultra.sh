set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
ULTRAGROTH_TRACE=1 timeout -k 10 600 python bench.py --ultra --log-domain 22 --steps 3 --warmup 1 --check > gpurun_out/r2_ultra22.json 2> gpurun_out/r2_ultra22.err; rc=$?
grep "ultragroth\]" gpurun_out/r2_ultra22.err | tail -14
tail -2 gpurun_out/r2_ultra22.err | cut -c1-300
cat gpurun_out/r2_ultra22.json | cut -c1-900
exit $rc

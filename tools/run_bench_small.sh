set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python bench.py --log-domain 16 --steps 2 --warmup 1 --check --cpu-sample-log 14 > gpurun_out/bench16.json 2> gpurun_out/bench16.err; echo "rc=$?"; tail -3 gpurun_out/bench16.err; cat gpurun_out/bench16.json
timeout -k 10 600 python bench.py --log-domain 20 --steps 3 --warmup 1 --cpu-sample-log 18 > gpurun_out/bench20.json 2> gpurun_out/bench20.err; echo "rc=$?"; tail -3 gpurun_out/bench20.err; cat gpurun_out/bench20.json

# round 3: bench.py --gpus N --check rehearsed over gloo on one GPU for every step order the launcher can take: chain first
# (what 8 ranks do), chain beside the products, the blocking phase calls; and a world that does not split the domain evenly
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=8
export UG_BENCH_BACKEND=gloo UG_BENCH_ONE_DEVICE=1
run() {   # N, extra env (as VAR=VALUE words)
  local N=$1; shift
  env "$@" timeout -k 10 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus $N --steps 2 --warmup 1 --log-domain 18 --no-cpu-baseline --check > gpurun_out/mo.json 2> gpurun_out/mo.err
  local rc=$?
  python -c "
import json; d=json.loads(open('gpurun_out/mo.json').read().strip().splitlines()[-1]); print('N=$N $* rc=$rc', d['n_gpus'], '%.2f ms' % d['ms_per_step'], d['config']['parallelism'], d.get('check'))" || { tail -5 gpurun_out/mo.err | cut -c1-300; return 1; }
  return $rc
}
run 4 UG_BENCH_CHAIN_ORDER=first && run 4 UG_BENCH_CHAIN_ORDER=beside && run 2 UG_BENCH_CHAIN_ORDER=first && run 4 UG_BENCH_PHASES=sequential && run 3 UG_BENCH_CHAIN_ORDER=auto && run 5 UG_BENCH_CHAIN_ORDER=auto

set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=8
export UG_BENCH_BACKEND=gloo UG_BENCH_ONE_DEVICE=1
for N in 2 4; do
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --steps 2 --warmup 1 --log-domain 18 --no-cpu-baseline --check > gpurun_out/multi$N.json 2> gpurun_out/multi$N.err; echo "N=$N rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" gpurun_out/multi$N.err | tail -5 | cut -c1-300; python -c "
import json; d=json.loads(open('gpurun_out/multi$N.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['config']['parallelism'], d.get('check'))"
done
# --replicas: the extra throughput figure (every rank a whole prover of its own)
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 2 --steps 2 --warmup 1 --log-domain 18 --no-cpu-baseline --check --replicas > gpurun_out/multirep.json 2> gpurun_out/multirep.err; echo "replicas rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/multirep.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d.get('check'), d['replicated'])"
# the bucket-class layouts (ULTRAGROTH_SHARD=PxB), four and five ranks
for SH in 1x4 2x2 1x5; do
N=${SH#*x}; P=${SH%x*}; N=$((N * P))
ULTRAGROTH_SHARD=$SH timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus $N --steps 2 --warmup 1 --log-domain 18 --no-cpu-baseline --check > gpurun_out/multic$SH.json 2> gpurun_out/multic$SH.err; echo "classes $SH rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" gpurun_out/multic$SH.err | tail -5 | cut -c1-300; python -c "
import json; d=json.loads(open('gpurun_out/multic$SH.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['config']['parallelism'], d.get('check'))"
done
for N in 2 3; do
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus $N --steps 2 --warmup 1 --log-domain 14 --ultra --check > gpurun_out/multiu$N.json 2> gpurun_out/multiu$N.err; echo "ultra N=$N rc=$?"; grep -v "socket.cpp\|Gloo\|amdgpu.ids" gpurun_out/multiu$N.err | tail -5 | cut -c1-300; python -c "
import json; d=json.loads(open('gpurun_out/multiu$N.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['ms_per_step'], d['config']['parallelism'], d['config']['workload'][-20:])"
done
# the driver's line at its own size: four ranks at 2^24 on the one device (32 s wall; the time per step means nothing here)
( time timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29661 bench.py --gpus 4 --steps 3 --warmup 1 --check ) > gpurun_out/multi4_full.json 2> gpurun_out/multi4_full.err; echo "2^24 N=4 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/multi4_full.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['config']['parallelism'], d.get('check'), 'create', d['create_s'])"

#!/bin/bash
# round 3: the G2 accumulation's table gathers as non-temporal loads (UG_NT_GATHER=1, default) against plain loads (=0), same box, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for nt in 1 0 1 0; do echo "UG_NT_GATHER=$nt"; UG_NT_GATHER=$nt timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --host-threads 1 --check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ks=dict(r['kernels']); ks[r['kernel']]=r
print('  ms/step %.2f msm %.2f | ' % (d['ms_per_step'], d['msm_ms_per_proof']) + ' | '.join('%s %.2f' % (k[-22:], v['avg_launch_ms']) for k, v in ks.items()), d.get('check'))"; done

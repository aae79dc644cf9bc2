#!/bin/bash
# round-end rehearsal of what the driver runs: the whole GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r4_final_suite.log 2>&1; echo "suite rc=$?"; tail -16 gpurun_out/r4_final_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r4_final_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.3f proofs/s  %.2f ms/step  msm %.1f | fft %.1f  prove call %.2f ms  pipelined %.2f/s  create %.2f s' % (d['value'], d['ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof'], d['prove_call_ms_per_step'], d['pipelined_proofs_per_s'], d['create_s']))
print('roofline', r['kernel'], 'launch %.2f ms  frac %.4f  issue %.3f' % (r['avg_launch_ms'], r['frac'], r['issue_bound']['frac']), 'traffic', r['traffic'])
print('cpu', d['cpu_baseline']['seconds_per_proof'], d['cpu_baseline']['cores'])"

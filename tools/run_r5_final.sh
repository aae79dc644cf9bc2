#!/bin/bash
# round-end rehearsal of what the driver runs: the whole GPU suite, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r5_final_suite.log 2>&1; echo "suite rc=$?"; tail -16 gpurun_out/r5_final_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
S=$(date +%s)
python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python -c "
import json; d=json.loads(open('gpurun_out/r5_final_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.3f proofs/s  %.2f ms/step (un-overlapped %.2f)  msm %.1f | fft %.1f  api %.2f ms = %.3f /s  pipelined %.2f/s  create %.2f s  first proof %.2f s  tables %.2f s' % (d['value'], d['ms_per_step'], d['unoverlapped_ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof'], d['api_ms_per_step'], d['api_value'], d['pipelined_proofs_per_s'], d['create_s'], d['time_to_first_proof_s'], d['tables_in_use_after_s']))
print('roofline', r['kernel'], r['bound'], 'launch %.2f ms  hbm frac %.4f  issue %.3f' % (r['avg_launch_ms'], r['frac'], r['issue_bound']['frac']), 'traffic', r['traffic'], r['traffic_source'][:40])
print('board', r['board'])
print('cpu', d['cpu_baseline']['seconds_per_proof'], d['cpu_baseline']['cores'])"

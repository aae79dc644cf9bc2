#!/bin/bash
# round 3, end of round: the whole -m gpu suite, smoke(), the driver's default bench line, the sharded code path over a one-rank RCCL group at 2^24
set -o pipefail
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r3_final_suite.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_final_suite.log; tail -8 gpurun_out/r3_final_suite.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err; tail -4 gpurun_out/r3_final_bench.err
UG_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29677 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --check > gpurun_out/r3_final_forced_dist.json 2> gpurun_out/r3_final_forced_dist.err
python - <<'PY'
import json
for f in ("gpurun_out/r3_final_bench.json", "gpurun_out/r3_final_forced_dist.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "msm_ms_per_proof", "fft_ms_per_proof", "prove_call_ms_per_step", "pipelined_proofs_per_s", "check", "comm")})
PY

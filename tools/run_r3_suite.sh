#!/bin/bash
# round 3: GPU suite without the full-size file, then the driver's default bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_registry.py tests/test_gpu_faults.py tests/test_trapdoor.py -x -q -m gpu > gpurun_out/r3_t4.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_t4.log; tail -15 gpurun_out/r3_t4.log
[ $rc -ne 0 ] && exit $rc
( time python bench.py "$@" ) > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"; tail -4 gpurun_out/r3_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "msm_ms_per_proof", "fft_ms_per_proof", "prove_call_ms_per_step", "witness_upload_ms_per_proof", "pipelined_proofs_per_s", "create_s")})
r = d["roofline"]; print(r["kernel"], r["avg_launch_ms"], r["launches"], r["frac"], r.get("issue_bound"))
for k, v in r["kernels"].items(): print("  ", k, v["avg_launch_ms"], v["launches"], round(v["frac"], 4))
print(d.get("cpu_baseline"))
PY

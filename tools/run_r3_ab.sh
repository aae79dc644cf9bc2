#!/bin/bash
# round 3: parity of the fused A|B1|C group path, then whole-library A/B at 2^24 (fused vs separate sets, both kernel shapes)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_registry.py tests/test_gpu_faults.py tests/test_trapdoor.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_t2.log; tail -3 gpurun_out/r3_t2.log
[ $rc -ne 0 ] && exit $rc
for cfg in "1 0" "1 1" "0 0"; do
  set -- $cfg
  echo "== FUSED=$1 ROTATE=$2" >> gpurun_out/r3_ab.log
  ULTRAGROTH_FUSED=$1 UG_GROUP_ROTATE=$2 timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --check >> gpurun_out/r3_ab.log 2>gpurun_out/r3_ab.err || { echo "bench failed" >> gpurun_out/r3_ab.log; tail -5 gpurun_out/r3_ab.err; exit 1; }
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3_ab.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln); k = d["roofline"]["kernels"]
    print("  ms/step %.2f seq %.2f msm %.2f fft %.2f | g1 %.2f x%d | grp %.2f x%d | g2 %.2f | ntt %.3f | %s" % (
        d["ms_per_step"], d["sequential_ms_per_step"], d["msm_ms_per_proof"], d["fft_ms_per_proof"],
        d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], k["segment_accumulate_group_kernel<3>"]["avg_launch_ms"],
        k["segment_accumulate_group_kernel<3>"]["launches"], k["segment_accumulate_kernel<G2Cfg>"]["avg_launch_ms"],
        k["ntt_pass_kernel"]["avg_launch_ms"], d.get("check")))
PY

#!/bin/bash
# (round 4: the A/B switches this recipe sets exist only in the -DUG_MEASURE build -- make -C ultragroth_amd/csrc MEASURE=1 measure)
export ULTRAGROTH_LIB=${GRAFT_REPO_ROOT:-$PWD}/ultragroth_amd/csrc/libultragroth_hip_measure.so
# round 3: the hand-written radix partition against the library sort (UG_SORT=cub), whole-library A/B at 2^24; then the gather ablation
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_registry.py tests/test_gpu_faults.py tests/test_trapdoor.py -x -q -m gpu > gpurun_out/r3_t5.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_t5.log; tail -12 gpurun_out/r3_t5.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r3_sort.log
for cfg in "own" "cub"; do
  echo "== SORT=$cfg" >> gpurun_out/r3_sort.log
  UG_SORT=$cfg timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --check >> gpurun_out/r3_sort.log 2>gpurun_out/r3_sort.err || { echo "bench failed" >> gpurun_out/r3_sort.log; tail -5 gpurun_out/r3_sort.err; exit 1; }
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3_sort.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln)
    print("  ms/step %.2f msm %.2f fft %.2f call %.2f pipelined %.3f/s | %s" % (d["ms_per_step"], d["msm_ms_per_proof"], d["fft_ms_per_proof"],
          d["prove_call_ms_per_step"], d["pipelined_proofs_per_s"], d.get("check")))
PY
bash tools/run_r3_fold.sh

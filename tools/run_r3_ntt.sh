#!/bin/bash
# (round 4: the A/B switches this recipe sets exist only in the -DUG_MEASURE build -- make -C ultragroth_amd/csrc MEASURE=1 measure)
export ULTRAGROTH_LIB=${GRAFT_REPO_ROOT:-$PWD}/ultragroth_amd/csrc/libultragroth_hip_measure.so
# round 3: NTT with unpacked twiddles and the three chains batched per pass; mat-vec with four entries in flight
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r3_t3.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_t3.log; tail -3 gpurun_out/r3_t3.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r3_ntt.log
for cfg in "1" "0"; do
  echo "== NTT_BATCH=$cfg" >> gpurun_out/r3_ntt.log
  UG_NTT_BATCH=$cfg timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --check >> gpurun_out/r3_ntt.log 2>gpurun_out/r3_ntt.err || { echo "bench failed" >> gpurun_out/r3_ntt.log; tail -5 gpurun_out/r3_ntt.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof -o r3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/r3_prof.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, glob, csv
for ln in open("gpurun_out/r3_ntt.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln); k = d["roofline"]["kernels"]
    print("  ms/step %.2f seq %.2f msm %.2f fft %.2f | ntt %.3f x%d | create %.2f | %s" % (
        d["ms_per_step"], d["sequential_ms_per_step"], d["msm_ms_per_proof"], d["fft_ms_per_proof"],
        k["ntt_pass_kernel"]["avg_launch_ms"], k["ntt_pass_kernel"]["launches"], d["create_s"], d.get("check")))
for f in glob.glob("gpurun_out/r3_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print("%-70s calls %5s total %9.3f ms avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY

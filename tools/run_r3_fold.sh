#!/bin/bash
# round 3 measurement: what the fused A|B1|C accumulation would take if its gathers never missed (indices folded into cache)
set -o pipefail
mkdir -p gpurun_out; rm -f gpurun_out/r3_fold.log
for f in ${FOLDS:-"" 14 18}; do
  echo "== FOLD_LOG=$f" >> gpurun_out/r3_fold.log
  UG_GROUP_FOLD_LOG=$f timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 >> gpurun_out/r3_fold.log 2>gpurun_out/r3_fold.err || { echo failed; tail -3 gpurun_out/r3_fold.err; exit 1; }
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3_fold.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln); r = d["roofline"]; ks = dict(r["kernels"]); ks[r["kernel"]] = r
    print("  ms/step %.2f call %.2f | " % (d["ms_per_step"], d["prove_call_ms_per_step"]) + " | ".join("%s %.2f" % (k.split("<")[0][-14:] + k[-4:], v["avg_launch_ms"]) for k, v in ks.items()))
PY

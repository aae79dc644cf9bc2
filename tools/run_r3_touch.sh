#!/bin/bash
# round 3: group accumulation with an L2 touch of the records two / three turns ahead (UG_GROUP_TOUCH) against the plain rotation
set -o pipefail
mkdir -p gpurun_out; rm -f gpurun_out/r3_touch.log
for cfg in 0 2 3 0 2; do
  echo "== GROUP_TOUCH=$cfg" >> gpurun_out/r3_touch.log
  UG_GROUP_TOUCH=$cfg timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --host-threads 1 --check >> gpurun_out/r3_touch.log 2>gpurun_out/r3_touch.err || { echo "bench failed"; tail -5 gpurun_out/r3_touch.err; exit 1; }
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3_touch.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln); r = d["roofline"]; ks = dict(r["kernels"]); ks[r["kernel"]] = r
    print("  ms/step %.2f msm %.2f | " % (d["ms_per_step"], d["msm_ms_per_proof"]) + " | ".join("%s %.2f" % (k[-22:], v["avg_launch_ms"]) for k, v in ks.items()) + " | " + str(d.get("check")))
PY

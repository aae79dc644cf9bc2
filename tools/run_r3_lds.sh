#!/bin/bash
# round 3: group accumulation with the K records of an entry fetched together by LDS-DMA (UG_GROUP_LDS=1) against the rotation
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "group or batch or window_tables" 2>&1 | tail -3 || exit 1
UG_GROUP_LDS=1 timeout -k 10 300 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "irregular or synthetic or sharded or ultragroth_matches or created_prover" 2>&1 | tail -3 || exit 1
rm -f gpurun_out/r3_lds.log
for cfg in 0 1 0 1; do
  echo "== GROUP_LDS=$cfg" >> gpurun_out/r3_lds.log
  UG_GROUP_LDS=$cfg timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --host-threads 1 --check >> gpurun_out/r3_lds.log 2>gpurun_out/r3_lds.err || { echo "bench failed"; tail -5 gpurun_out/r3_lds.err; exit 1; }
done
python - <<'PY'
import json
for ln in open("gpurun_out/r3_lds.log"):
    if ln.startswith("=="): print(ln.strip()); continue
    if not ln.startswith("{"): continue
    d = json.loads(ln); r = d["roofline"]; ks = dict(r["kernels"]); ks[r["kernel"]] = r
    print("  ms/step %.2f msm %.2f | " % (d["ms_per_step"], d["msm_ms_per_proof"]) + " | ".join("%s %.2f" % (k[-22:], v["avg_launch_ms"]) for k, v in ks.items()) + " | " + str(d.get("check")))
PY

#!/bin/bash
# (round 4: the A/B switches this recipe sets exist only in the -DUG_MEASURE build -- make -C ultragroth_amd/csrc MEASURE=1 measure)
export ULTRAGROTH_LIB=${GRAFT_REPO_ROOT:-$PWD}/ultragroth_amd/csrc/libultragroth_hip_measure.so
# round 3: tile shapes of the radix partition (pairs per lane in the pair-form passes, scalars per lane in the fused first pass)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
IFS=";" read -ra CFGS <<< "${UG_CFGS:-16 4;16 8;16 16}"; for cfg in "${CFGS[@]}"; do
  set -- $cfg
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/sk
  UG_SORT_IPT=$1 UG_SORT_LBW=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-threads 1 --check > $GRAFT_REPO_ROOT/gpurun_out/sk.json 2> $GRAFT_REPO_ROOT/gpurun_out/sk.err || { echo "failed $cfg"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/sk.err; exit 1; }
  python3 - "$cfg" <<'PY'
import csv, glob, json, sys, os
root = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(root + '/gpurun_out/sk/*/*kernel_stats.csv')[0]
rows = {r['Name']: r for r in csv.DictReader(open(f))}
d = json.loads(open(root + '/gpurun_out/sk.json').read().strip().splitlines()[-1])
out = "IPT LBW = %s: ms/step %.2f check %s |" % (sys.argv[1], d["ms_per_step"], d.get("check"))
for name, r in rows.items():
    if "radix" in name: out += " %s %.3f ms x%s |" % (name.split("(")[0].replace("ug::(anonymous namespace)::", "").replace("void ", ""), float(r["AverageNs"]) / 1e6, r["Calls"])
print(out)
PY
done

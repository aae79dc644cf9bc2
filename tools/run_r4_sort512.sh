#!/bin/bash
# Round 4 A/B: the radix partition with 512-lane workgroups (tiles of 8 192 pairs, 256-byte runs per bin and tile instead of
# 128-byte runs): ultragroth_amd/csrc/build/variants/libug_sort512.so (sort.hip built with -DUG_SORT_THREADS=512 and linked with
# the other objects of build/) against the product library, under rocprofv3 kernel statistics on ONE box; --check first.
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/ultragroth_amd/csrc/build/variants/libug_${CHECK_VARIANT:-sort512}.so
ULTRAGROTH_LIB=$V python bench.py --log-domain 20 --steps 1 --warmup 1 --no-cpu-baseline --check > gpurun_out/s512_check.json 2> gpurun_out/s512_check.err; echo "variant --check 2^20 U rc=$?"
ULTRAGROTH_LIB=$V python bench.py --log-domain 20 --mix C --steps 1 --warmup 1 --no-cpu-baseline --check > gpurun_out/s512_checkc.json 2> gpurun_out/s512_checkc.err; echo "variant --check 2^20 C rc=$?"
cd /tmp && export TMPDIR=/tmp
for WHICH in ${VARIANTS:-base sort512 base sort512}; do
  L=""; [ $WHICH != base ] && L=$GRAFT_REPO_ROOT/ultragroth_amd/csrc/build/variants/libug_$WHICH.so
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ab_$WHICH
  ULTRAGROTH_LIB=$L timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_$WHICH -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain 24 --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > $GRAFT_REPO_ROOT/gpurun_out/ab_$WHICH.json 2> $GRAFT_REPO_ROOT/gpurun_out/ab_$WHICH.err
  python3 - $WHICH <<'PY'
import csv, glob, json, sys, os
w = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(glob.glob(root + "/gpurun_out/ab_%s/*/*kernel_stats.csv" % w)[0])))
d = json.loads(open(root + "/gpurun_out/ab_%s.json" % w).read().strip().splitlines()[-1])
pick = {r["Name"].replace("ug::(anonymous namespace)::", "").replace("void ", "").split("(")[0]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in rows}
print(w, "ms/step %.2f msm %.2f | " % (d["ms_per_step"], d["msm_ms_per_proof"]) + "  ".join("%s %d x %.1f us" % (k, v[0], v[1]) for k, v in pick.items() if "radix" in k or "transpose" in k))
PY
done

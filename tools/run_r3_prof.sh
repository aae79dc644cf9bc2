#!/bin/bash
# round 3 evidence: rocprofv3 kernel statistics of the default bench command, then FETCH_SIZE / WRITE_SIZE in separate passes, then
# the lane-pair G2 microbenchmark and the one-shot CLI time
set -o pipefail
L=${1:-24}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3prof -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > $GRAFT_REPO_ROOT/gpurun_out/r3prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3prof_bench.err; echo "stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3pmc_$C
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 2 --warmup 1 --no-cpu-baseline --host-threads 1 > $GRAFT_REPO_ROOT/gpurun_out/r3pmc_$C.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3pmc_$C.err; echo "$C rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob('gpurun_out/r3prof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
with open('gpurun_out/r3_kernel_stats_top.txt', 'w') as out:
    d = json.loads(open('gpurun_out/r3prof_bench.json').read().strip().splitlines()[-1])
    out.write("bench.py --log-domain 24 --steps 4 --warmup 1 --host-threads 1 under rocprofv3 --kernel-trace --stats: ms/step %.2f (resident), prove call %.2f ms\n" % (d["ms_per_step"], d["prove_call_ms_per_step"]))
    for r in rows[:40]:
        line = "%-78s calls=%5s total_ms=%9.3f avg_us=%10.1f" % (r['Name'].replace('ug::(anonymous namespace)::', '').replace('void ', '')[:78], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3)
        out.write(line + "\n")
print(open('gpurun_out/r3_kernel_stats_top.txt').read())
summary = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob('gpurun_out/r3pmc_%s/*/*counter_collection.csv' % c)
    if not fs:
        print("no counter file for", c); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('ug::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    print("==", c)
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%-60s calls=%4d  per-call KB=%14.1f" % (k[:60], n, v / n))
        summary.setdefault(k, {})[c] = v / n
json.dump(summary, open('gpurun_out/r3_pmc_raw.json', 'w'), indent=1)
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iultragroth_amd/csrc tools/ubench_g2pair.hip -o tools/ubench_g2pair 2>/dev/null
timeout -k 10 120 ./tools/ubench_g2pair | tee gpurun_out/r3_ubench_g2pair.txt
timeout -k 10 300 ./tools/ubench_madd | tee gpurun_out/r3_ubench_madd.txt
timeout -k 10 600 python tools/oneshot_cli.py $L | tee gpurun_out/r3_oneshot.json

#!/bin/bash
# Round 4 evidence in one call: rocprofv3 kernel statistics of the default bench command; FETCH_SIZE / WRITE_SIZE in separate
# counter passes; their summary (tools/pmc_summary.py); then the bench line itself with `roofline.traffic` taken from THOSE
# passes (--pmc-summary: measured in this very call, not read from a committed file).
set -o pipefail
L=${1:-24}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r4prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4prof -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > $O/r4prof_bench.json 2> $O/r4prof_bench.err; echo "stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/r4pmc_$C
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/r4pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 2 --warmup 1 --no-cpu-baseline --host-threads 1 > $O/r4pmc_$C.json 2> $O/r4pmc_$C.err; echo "$C rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/r4pmc_FETCH_SIZE $O/r4pmc_WRITE_SIZE $O/r04_pmc_summary.json $L
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/r4prof/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
with open('gpurun_out/r04_kernel_stats_top.txt', 'w') as out:
    d = json.loads(open('gpurun_out/r4prof_bench.json').read().strip().splitlines()[-1])
    out.write("bench.py --log-domain %d --steps 4 --warmup 1 --host-threads 1 under rocprofv3 --kernel-trace --stats: ms/step %.2f (resident), prove call %.2f ms\n" % (d["config"]["log_domain"], d["ms_per_step"], d["prove_call_ms_per_step"]))
    for r in rows[:40]:
        out.write("%-78s calls=%5s total_ms=%9.3f avg_us=%10.1f\n" % (r['Name'].replace('ug::(anonymous namespace)::', '').replace('void ', '')[:78], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print(open('gpurun_out/r04_kernel_stats_top.txt').read())
PY
cp gpurun_out/r4prof/*/*kernel_stats.csv gpurun_out/r04_bench24_kernel_stats.csv
python3 bench.py --log-domain $L --steps 8 --warmup 1 --pmc-summary gpurun_out/r04_pmc_summary.json > gpurun_out/r04_bench_with_pmc.json 2> gpurun_out/r04_bench_with_pmc.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04_bench_with_pmc.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['prove_call_ms_per_step'], r['kernel'], r['frac'], r['traffic'], r['traffic_source'], d['cpu_baseline']['value'])"

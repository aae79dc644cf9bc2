#!/bin/bash
# Round 5 evidence in one call: (1) rocprofv3 kernel statistics of the bench command with the H branch BEHIND the products
# (--overlap 0: one kernel on the chip at a time, the launch times the roofline is computed from); (2) the same under the default
# (--overlap 1) for the record; (3) the default `python bench.py` itself, whose counter traffic is measured by its own child runs
# (the summary they produce is kept: UG_BENCH_PMC_DUMP).
set -o pipefail
L=${1:-24}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for OV in 0 1; do
  rm -rf $O/r5prof_ov$OV
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5prof_ov$OV -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --host-threads 1 --overlap $OV > $O/r5prof_ov${OV}_bench.json 2> $O/r5prof_ov${OV}_bench.err; echo "stats overlap=$OV rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json
for ov in (0, 1):
    f = glob.glob('gpurun_out/r5prof_ov%d/*/*kernel_stats.csv' % ov)[0]
    rows = list(csv.DictReader(open(f)))
    d = json.loads(open('gpurun_out/r5prof_ov%d_bench.json' % ov).read().strip().splitlines()[-1])
    with open('gpurun_out/r05_kernel_stats_top_ov%d.txt' % ov, 'w') as out:
        out.write("bench.py --log-domain %d --steps 4 --warmup 1 --host-threads 1 --overlap %d --no-pmc under rocprofv3 --kernel-trace --stats: ms/step %.2f (resident), "
                  "un-overlapped %s, groth16_prover_prove %.2f ms; the line's launch times: group %.3f G2 %.3f H %.3f NTT %.4f ms\n" % (
                      d["config"]["log_domain"], ov, d["ms_per_step"], d.get("unoverlapped_ms_per_step"), d["api_ms_per_step"],
                      *[({**d["roofline"]["kernels"], d["roofline"]["kernel"]: d["roofline"]}[k]["avg_launch_ms"]) for k in
                        ("segment_accumulate_group_kernel<3>", "segment_accumulate_kernel<G2Cfg>", "segment_accumulate_kernel<G1Cfg>", "ntt_pass_kernel")]))
        for r in rows[:42]:
            out.write("%-78s calls=%5s total_ms=%9.3f avg_us=%10.1f\n" % (r['Name'].replace('ug::(anonymous namespace)::', '').replace('void ', '')[:78], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
    print(open('gpurun_out/r05_kernel_stats_top_ov%d.txt' % ov).read()[:2400])
PY
cp gpurun_out/r5prof_ov0/*/*kernel_stats.csv gpurun_out/r05_bench24_kernel_stats_overlap0.csv
cp gpurun_out/r5prof_ov1/*/*kernel_stats.csv gpurun_out/r05_bench24_kernel_stats_overlap1.csv
S=$(date +%s)
UG_BENCH_PMC_DUMP=$O/r05_pmc_summary.json python3 bench.py > gpurun_out/r05_bench24_default.json 2> gpurun_out/r05_bench24_default.err; echo "bench rc=$? wall=$(( $(date +%s) - S )) s"
python3 -c "
import json; d=json.loads(open('gpurun_out/r05_bench24_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.3f  ms/step %.2f  unoverlapped %.2f  api %.2f ms (%.3f /s)  create %.2f first proof %.2f tables %.2f' % (d['value'], d['ms_per_step'], d['unoverlapped_ms_per_step'], d['api_ms_per_step'], d['api_value'], d['create_s'], d['time_to_first_proof_s'], d['tables_in_use_after_s']))
print(r['kernel'], r['bound'], r['bound_frac'], r['frac'], r['avg_launch_ms'], r['traffic'], r['traffic_source'][:60]); print(r['board'])"

# kernel statistics of bench.py at 2^L with the H branch BEHIND the products (--overlap 0: one kernel on the chip at a time)
set -o pipefail
L=${1:-22}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_clean
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_clean -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 8 --warmup 2 --bare --overlap 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_clean_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_clean_bench.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/prof_clean/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
d=json.load(open('gpurun_out/prof_clean_bench.json'))
print("ms/step %.3f  msm %.2f  fft %.2f  (10 proofs on the tables + 1 on classic windows in the counts)" % (d['ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof']))
for r in rows[:34]:
    n=r['Name'].replace('ug::(anonymous namespace)::','').replace('void ','')
    if 'window_tables' in n or 'synth_points' in n: continue
    print("%-62s calls=%4s total_ms=%9.3f avg_us=%9.1f  per proof %.3f ms" % (n[:62], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/11))
PY

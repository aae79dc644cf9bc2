set -o pipefail
L=${1:-22}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-62s calls=%4s total_ms=%9.3f avg_us=%9.1f" % (r['Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:62], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
d=json.load(open('gpurun_out/prof_bench.json'))
print(d['ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof'])
PY

# kernel stats of the circom-like scalar mix at 2^L: bash tools/run_prof_c.sh L
set -o pipefail
L=${1:-24}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --mix C --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_c.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_c.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/prof_c/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    print("%-62s calls=%4s total_ms=%9.3f avg_us=%9.1f" % (r['Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:62], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
d=json.load(open('gpurun_out/prof_c.json'))
print(d['ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof'])
PY

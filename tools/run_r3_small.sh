# round 3: per-entry rate of the group accumulation at small sizes -- one GPU alone (bench.py at 2^21) against a rank of an 8-way
# shard of 2^24 (tools/phase_times.py under rocprofv3), and the segment length of that rank: bash tools/run_r3_small.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --log-domain 21 --steps 10 --warmup 2 --no-cpu-baseline --host-threads 1 > gpurun_out/small21.json 2> gpurun_out/small21.err || exit 1
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/small21.json').read().strip().splitlines()[-1])
r = d['roofline']
print('2^21 alone: step %.2f ms' % d['ms_per_step'], r['kernel'], '%.3f ms' % r['avg_launch_ms'], {k: round(v['avg_launch_ms'], 3) for k, v in r['kernels'].items()})
PY
for L in 20 18 17; do
  echo "rank 5 of 8, UG_SEG_LANES_LOG=$L"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_small
  UG_SEG_LANES_LOG=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -- python3 $GRAFT_REPO_ROOT/tools/phase_times.py 24 8 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_small_$L.txt 2>/dev/null || exit 1
  cd $GRAFT_REPO_ROOT
  tail -n 3 gpurun_out/prof_small_$L.txt | cut -c1-220
  python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_small/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'segment_accumulate' in n or 'bucket_fixup' in n or 'bucket_chunk' in n or 'medium' in n:
        print("   %-60s calls=%3s avg_us=%9.1f" % (n.replace('ug::(anonymous namespace)::', '').replace('void ', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done

# kernel stats of ONE rank of a W-way shard (tools/phase_times.py): bash tools/run_prof_rank.sh [log] [world] [rank] [point ranges] [mix]
set -o pipefail
L=${1:-24}; W=${2:-8}; R=${3:-5}; P=${4:-$W}; MIX=${5:-U}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_rank
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rank -- python3 $GRAFT_REPO_ROOT/tools/phase_times.py $L $W $R 9 products_first $P $MIX > $GRAFT_REPO_ROOT/gpurun_out/prof_rank.txt 2> $GRAFT_REPO_ROOT/gpurun_out/prof_rank.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_rank.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_rank/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:34]:
    print("%-70s calls=%4s total_ms=%9.3f avg_us=%9.1f" % (r['Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY

# instruction-cache and issue counters of the accumulation kernels, default segments against short ones:
#   bash tools/run_pmc_icache.sh [log]
set -o pipefail
L=${1:-24}
for lanes in 20 22; do
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_ic
UG_SEG_TAPER=0 UG_SEG_LANES_LOG=$lanes timeout -k 10 500 rocprofv3 --pmc ${UG_PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES} --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ic -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 1 --warmup 1 --host-threads 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_ic.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_ic.err; echo "UG_SEG_LANES_LOG=$lanes rc=$?"; tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_ic.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob('gpurun_out/pmc_ic/*/*counter_collection.csv')
agg=collections.defaultdict(lambda:collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
names=sorted({n for v in agg.values() for n in v})
print("%-42s"%"kernel"+" ".join("%16s"%n for n in names))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0))[:4]:
    print("%-42s"%k+" ".join("%16.4g"%(v[n]) for n in names))
PY
done

set -o pipefail
L=${1:-22}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.err; echo "rc=$?"; tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv')
agg=collections.defaultdict(lambda:collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
names=['SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_WAIT_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_INSTS_VALU']
print("%-42s"%"kernel"+" ".join("%14s"%n[3:] for n in names))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]['SQ_WAVE_CYCLES'])[:12]:
    wc=v['SQ_WAVE_CYCLES'] or 1
    print("%-42s"%k+" ".join("%14.3g"%(v[n]) for n in names))
    print("%-42s"%"   (share of wave cycles)"+" ".join("%14.2f"%(v[n]/wc) for n in names))
PY

set -o pipefail
L=${1:-22}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain $L --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.err; echo "$C rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv'%c)
    if not fs: print("no counter file for",c, glob.glob('gpurun_out/pmc_%s/*/*'%c)); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].replace('ug::(anonymous namespace)::','').replace('void ','')[:48]
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    print("==",c)
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
        print("%-50s calls=%4d  per-call=%12.1f" % (k,n,v/n))
PY

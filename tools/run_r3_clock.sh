#!/bin/bash
# round 3: effective shader clock per kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md "DVFS give-back") of the bench step
set -o pipefail
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3clk
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3clk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-threads 1 > $GRAFT_REPO_ROOT/gpurun_out/r3clk.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3clk.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob('gpurun_out/r3clk/*/*counter_collection.csv')[0]
kt = glob.glob('gpurun_out/r3clk/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp']), r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d = dur.get(r['Dispatch_Id'])
    if not d: continue
    k = d[1].replace('ug::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    agg[k][0] += 1; agg[k][1] += float(r['Counter_Value']); agg[k][2] += d[0]
out = []
for k, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:14]:
    out.append("%-58s calls=%4d avg %9.3f ms  effective clock %.3f GHz" % (k[:58], n, ns / n / 1e6, cyc / 8.0 / ns))
print("\n".join(out))
open('gpurun_out/r3_clock_per_kernel.txt', 'w').write("effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace on bench.py --log-domain 24)\n" + "\n".join(out) + "\n")
PY
./tools/ubench_clock 2>&1 | tail -8 | tee -a gpurun_out/r3_clock_per_kernel.txt

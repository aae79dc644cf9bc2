# round 3: the radix partition without the zero digits (drop mode of RadixSorter::sort, UG_SORT_DROP=0 for the old form): the parity
# (round 4: the A/B switches this recipe sets exist only in the -DUG_MEASURE build -- make -C ultragroth_amd/csrc MEASURE=1 measure)
export ULTRAGROTH_LIB=${GRAFT_REPO_ROOT:-$PWD}/ultragroth_amd/csrc/libultragroth_hip_measure.so
# suites, then both forms at 2^24 on the circom-like and the uniform mix with --check, radix kernels under rocprofv3:
# bash tools/run_r3_drop.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_trapdoor.py tests/test_gpu_registry.py tests/test_gpu_faults.py -m gpu -x -q > gpurun_out/drop_tests.log 2>&1 || { tail -20 gpurun_out/drop_tests.log; exit 1; }
tail -n 2 gpurun_out/drop_tests.log
for cfg in "C 1" "C 0" "U 1" "U 0" "C 1" "C 0"; do
  set -- $cfg
  cd /tmp && export TMPDIR=/tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_drop
  UG_SORT_DROP=$2 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_drop -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain 24 --mix $1 --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 --check > $GRAFT_REPO_ROOT/gpurun_out/drop.json 2> $GRAFT_REPO_ROOT/gpurun_out/drop.err || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/drop.err; exit 1; }
  cd $GRAFT_REPO_ROOT
  python3 - "$1" "$2" <<'PY'
import csv, glob, json, sys
f = glob.glob('gpurun_out/prof_drop/*/*kernel_stats.csv')[0]
d = json.loads(open('gpurun_out/drop.json').read().strip().splitlines()[-1])
out = "mix %s drop %s: step %.2f ms  msm %.2f  check %s |" % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['msm_ms_per_proof'], d.get('check'))
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'radix_' in n or 'bucket_bounds' in n or 'transpose' in n:
        out += " %s %.0f us x%s |" % (n.replace('ug::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:22], float(r['AverageNs']) / 1e3, r['Calls'])
        tot += float(r['TotalDurationNs']) / 1e6
print(out + " sum %.1f ms" % tot)
PY
done

# round 3: coefficient mat-vec with the bit reversal in 16 x 16 LDS tiles (UG_MATVEC_TILED=1, default) against one scattered
# (round 4: the A/B switches this recipe sets exist only in the -DUG_MEASURE build -- make -C ultragroth_amd/csrc MEASURE=1 measure)
export ULTRAGROTH_LIB=${GRAFT_REPO_ROOT:-$PWD}/ultragroth_amd/csrc/libultragroth_hip_measure.so
# 32-byte store per lane (=0): parity of the H polynomial first, then both forms under rocprofv3 at 2^24: bash tools/run_r3_matvec.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_trapdoor.py -m gpu -x -q > gpurun_out/matvec_tests.log 2>&1 || { tail -20 gpurun_out/matvec_tests.log; exit 1; }
tail -n 2 gpurun_out/matvec_tests.log
for T in 1 0; do
  cd /tmp && export TMPDIR=/tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mv
  UG_MATVEC_TILED=$T timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mv -- python3 $GRAFT_REPO_ROOT/bench.py --log-domain 24 --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 --check > $GRAFT_REPO_ROOT/gpurun_out/mv_$T.json 2> $GRAFT_REPO_ROOT/gpurun_out/mv_$T.err || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/mv_$T.err; exit 1; }
  cd $GRAFT_REPO_ROOT
  python3 - $T <<'PY'
import csv, glob, json, sys
f = glob.glob('gpurun_out/prof_mv/*/*kernel_stats.csv')[0]
d = json.loads(open('gpurun_out/mv_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print("UG_MATVEC_TILED=%s: step %.2f ms  fft %.2f ms  check %s" % (sys.argv[1], d['ms_per_step'], d['fft_ms_per_proof'], d.get('check')))
for r in csv.DictReader(open(f)):
    if 'matvec' in r['Name'] or 'ntt_pass' in r['Name']:
        print("   %-50s calls=%3s avg_us=%9.1f" % (r['Name'].replace('ug::(anonymous namespace)::', '').replace('void ', '')[:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
# the NTT launches of the last proof of the last run, in order (three chains per launch: inverse passes 1-3, forward passes 1-3)
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_mv/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Grid_Size_Y']) for r in rows if 'ntt_pass' in r['Kernel_Name'] or 'matvec' in r['Kernel_Name']]
idx = [i for i, s in enumerate(seq) if 'matvec' in s[0]]
print("   last proof: " + "  ".join("%s%.3f(x%s)" % ("mv " if 'matvec' in s[0] else "", s[1], s[2]) for s in seq[idx[-2]:idx[-1]]))
PY

# Round 5: Shoup-form twiddle products in the NTT passes (UG_NTT_SHOUP=1, default) against the Montgomery form (=0): parity tests with
# both, then bench lines at 2^24 and 2^20 on one box (per-kernel launch times from the un-overlapped steps)
set -o pipefail
cd $GRAFT_REPO_ROOT
for S in 1 0; do
  UG_NTT_SHOUP=$S timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ntt or hpoly or fixture or field" > gpurun_out/r5_shoup_tests_$S.log 2>&1; echo "shoup=$S tests rc=$?"; tail -2 gpurun_out/r5_shoup_tests_$S.log
done
for rep in 1 2; do for S in 1 0; do
  UG_NTT_SHOUP=$S timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --check > gpurun_out/r5_shoup_b24_${S}_$rep.json 2> gpurun_out/r5_shoup_b24_${S}_$rep.err; echo "2^24 shoup=$S rc=$?"
done; done
for S in 1 0; do UG_NTT_SHOUP=$S timeout -k 10 300 python3 bench.py --log-domain 20 --steps 20 --warmup 2 --bare > gpurun_out/r5_shoup_b20_$S.json 2> gpurun_out/r5_shoup_b20_$S.err; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_shoup_b2*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
        k = dict(r['kernels']); k[r['kernel']] = r
        n = k.get('ntt_pass_kernel', {})
        print("%-28s ms/step %8.3f unoverlapped %8.3f fft %.3f  ntt launch %.4f ms x %d  issue %.3f  check %s" % (f.split('/')[-1], d['ms_per_step'], d['unoverlapped_ms_per_step'] or 0, d['fft_ms_per_proof'], n.get('avg_launch_ms', 0), n.get('launches', 0), n.get('issue_bound', {}).get('frac', 0), d.get('check')))
    except Exception as e:
        print(f, 'FAILED', e)
PY

# round 3: UltraGroth 2^22 (BASELINE configs[4] on one GPU) with the H branch ordered behind the witness products (default) and
# beside them (ULTRAGROTH_OVERLAP=1), alternating, --check: bash tools/run_r3_ultra_overlap.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
for ov in 0 1 0 1; do
  ULTRAGROTH_OVERLAP=$ov timeout -k 10 300 python bench.py --ultra --log-domain 22 --steps 6 --warmup 1 --host-threads 1 --check > gpurun_out/uo.json 2> gpurun_out/uo.err || { tail -3 gpurun_out/uo.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/uo.json').read().strip().splitlines()[-1]); print('ULTRAGROTH_OVERLAP=$ov: %.2f ms per call  msm %.2f  fft %.2f  %s' % (d['ms_per_step'], d['msm_ms_per_proof'], d['fft_ms_per_proof'], d['config']['workload'][-20:]))"
done

# Round 5: sparse B (signals without a B-side point): tests, then 2^24 with half of the B points at infinity, sparse against dense form
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5_sparse_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_sparse_tests.log
for Z in 0.5 0.75; do for S in 1 0; do
  ULTRAGROTH_SPARSE_B=$S timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --check --b-zero $Z > gpurun_out/r5_sparse_b24_z${Z}_s$S.json 2> gpurun_out/r5_sparse_b24_z${Z}_s$S.err; echo "b_zero=$Z sparse=$S rc=$?"
done; done
ULTRAGROTH_SPARSE_B=1 timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --check > gpurun_out/r5_sparse_b24_dense.json 2> gpurun_out/r5_sparse_b24_dense.err; echo "dense rc=$?"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_sparse_b24_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-34s ms/step %8.3f  unoverlapped %.3f  msm %.2f fft %.2f  create %.2f  %s" % (f.split('/')[-1], d["ms_per_step"], d["unoverlapped_ms_per_step"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], d["create_s"], d.get("check")))
    except Exception as e:
        print(f, "FAILED", e)
PY

# NTT / H-polynomial parity tests, then a quick 2^24 bench with the per-kernel split
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "ntt or hpoly or synthetic or tiny or irregular or sharded or ultragroth" > gpurun_out/r2_ntt_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r2_ntt_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --check "$@" > gpurun_out/r2_q24.json 2> gpurun_out/r2_q24.err; rc=$?
tail -3 gpurun_out/r2_q24.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2_q24.json").read().strip().splitlines()[-1])
r=d["roofline"]; k=r["kernels"]
print("%.1f ms/proof (%.2f proofs/s) upload %.1f | msm %.1f fft %.1f | G1 acc %.2f ms G2 acc %.2f ms ntt pass %.3f ms x %d | create %.2f s | check %s" % (
    d["ms_per_step"], d["value"], d["witness_upload_ms_per_proof"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], r["avg_launch_ms"],
    k["segment_accumulate_kernel<G2Cfg>"]["avg_launch_ms"], k["ntt_pass_kernel"]["avg_launch_ms"], k["ntt_pass_kernel"]["launches"], d["create_s"], d.get("check")))
PY
exit $rc

# quick 2^24 bench with the per-kernel split (no tests): bash tools/run_b24.sh [extra bench args]
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --log-domain 24 --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/q24.json 2> gpurun_out/q24.err || { tail -5 gpurun_out/q24.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/q24.json"))
r=d["roofline"]
k=r["kernels"]
print("seq %.1f ms | " % d.get("sequential_ms_per_step", 0) + "%.1f ms/proof (%.2f proofs/s)  upload %.1f  msm %.1f  fft %.1f | G1 acc %.2f ms  G2 acc %.2f ms  ntt pass %.3f ms" % (d["ms_per_step"], d["value"], d["witness_upload_ms_per_proof"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], r["avg_launch_ms"], k["segment_accumulate_kernel<G2Cfg>"]["avg_launch_ms"], k["ntt_pass_kernel"]["avg_launch_ms"]))
PY
python -c "
import json; d=json.load(open('gpurun_out/q24.json')); print('create_s %.2f' % d['create_s'])"

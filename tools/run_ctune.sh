set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for D in 4 5 6; do
for L in 20 22 24; do
UG_MSM_C_DELTA=$D timeout -k 10 600 python bench.py --log-domain $L --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/q$L.json 2> gpurun_out/q$L.err || tail -3 gpurun_out/q$L.err
python - <<PY
import json
d=json.load(open("gpurun_out/q$L.json"))
print("delta $D 2^$L: %.1f ms/proof  msm %.1f  fft %.1f  g1acc %.2f ms  g2acc %.2f ms" % (d["ms_per_step"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], d["roofline"]["avg_launch_ms"], d["roofline"]["g2_kernel"]["avg_launch_ms"]))
PY
done; done

set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in 20 22; do
timeout -k 10 400 python bench.py --log-domain $L --steps 5 --warmup 1 --no-cpu-baseline --check > gpurun_out/q$L.json 2> gpurun_out/q$L.err || { tail -5 gpurun_out/q$L.err; exit 1; }
python - $L <<'PY'
import json,sys
d=json.load(open("gpurun_out/q%s.json" % sys.argv[1]))
print("2^%s: %.2f ms/proof (%.2f proofs/s)  msm %.2f  fft %.2f  %s" % (sys.argv[1], d["ms_per_step"], d["value"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], d.get("check")))
PY
done

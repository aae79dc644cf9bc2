set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "%.1f ms/proof (%.2f proofs/s)  msm %.1f  fft %.1f" % (d["ms_per_step"], d["value"], d["msm_ms_per_proof"], d["fft_ms_per_proof"]))
PY
}
timeout -k 10 600 python bench.py --log-domain 22 --ultra --steps 3 --warmup 1 > gpurun_out/cfg4.json 2> gpurun_out/cfg4.err && show gpurun_out/cfg4.json || tail -3 gpurun_out/cfg4.err
timeout -k 10 600 python bench.py --log-domain 24 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/q24.json 2> gpurun_out/q24.err && show gpurun_out/q24.json || tail -3 gpurun_out/q24.err

set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "20" "20 --overlap" "22" "22 --overlap" "24" "24 --overlap" "20 --g1-only" "24 --mix C"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --log-domain "$@" --steps 5 --warmup 2 --no-cpu-baseline --check > gpurun_out/sz.json 2> gpurun_out/sz.err || { echo "FAILED $cfg"; tail -3 gpurun_out/sz.err; }
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sz.json").read().strip().splitlines()[-1])
print("%-16s %8.2f ms/proof with two host threads  %8.2f one after the other  upload %5.2f  msm %7.2f  fft %6.2f  check %s" % (sys.argv[1], d["ms_per_step"], d["sequential_ms_per_step"], d["witness_upload_ms_per_proof"], d["msm_ms_per_proof"], d["fft_ms_per_proof"], d.get("check")))
PY
done | tee gpurun_out/r2_sizes.txt

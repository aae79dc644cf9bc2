# window-width sweep of the tables mode: bash tools/run_tablec.sh "20 21 22" "18 19 20 21"
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in $1; do for C in $2; do
UG_TABLE_C=$C timeout -k 10 400 python bench.py --log-domain $L --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/tc.json 2> gpurun_out/tc.err || { tail -3 gpurun_out/tc.err; continue; }
python - $L $C <<'PY'
import json,sys
d=json.load(open("gpurun_out/tc.json"))
print("2^%s c=%s: %.2f ms/proof  msm %.2f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["msm_ms_per_proof"]))
PY
done; done

set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export OMP_NUM_THREADS=16
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "%.2f ms/proof (%.2f proofs/s)  unoverlapped %s  api %s  msm %.1f  fft %.1f  create %.2fs  first proof %s s" % (d["ms_per_step"], d["value"], d.get("unoverlapped_ms_per_step"), d.get("api_ms_per_step"), d["msm_ms_per_proof"], d["fft_ms_per_proof"], d["create_s"], d.get("time_to_first_proof_s")), d["config"]["workload"][:60])
PY
}
timeout -k 10 600 python bench.py --log-domain 20 --g1-only --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/cfg1.json 2> gpurun_out/cfg1.err && show gpurun_out/cfg1.json || tail -3 gpurun_out/cfg1.err
timeout -k 10 600 python bench.py --log-domain 24 --mix C --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/cfg2c.json 2> gpurun_out/cfg2c.err && show gpurun_out/cfg2c.json || tail -3 gpurun_out/cfg2c.err
timeout -k 10 600 python bench.py --log-domain 20 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/q20.json 2> gpurun_out/q20.err && show gpurun_out/q20.json || tail -3 gpurun_out/q20.err
timeout -k 10 600 python bench.py --log-domain 22 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/q22.json 2> gpurun_out/q22.err && show gpurun_out/q22.json || tail -3 gpurun_out/q22.err
timeout -k 10 600 python bench.py --log-domain 22 --ultra --steps 3 --warmup 1 > gpurun_out/cfg4.json 2> gpurun_out/cfg4.err && show gpurun_out/cfg4.json || tail -3 gpurun_out/cfg4.err
( time timeout -k 10 900 python bench.py --log-domain 26 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc ) > gpurun_out/cfg3.json 2> gpurun_out/cfg3.err && show gpurun_out/cfg3.json || tail -5 gpurun_out/cfg3.err
tail -4 gpurun_out/cfg3.err
# (round 4) the reference's largest legal domain, 2^27: the piecewise path (ranges above 2^26 scalars) at its own size
( time timeout -k 10 900 python bench.py --log-domain 27 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --host-threads 1 ) > gpurun_out/cfg27.json 2> gpurun_out/cfg27.err && show gpurun_out/cfg27.json || tail -5 gpurun_out/cfg27.err

# Round 5: the G1 and G2 tails of the witness products side by side (ug_msm_witness_enqueue, ULTRAGROTH_TAILS=split) against one product after the
# other (the default): tests, one rank of eight at 2^24, the full 2^24 and 2^20 problems, UltraGroth 2^22
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_faults.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r5_tails_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5_tails_tests.log
OUT=gpurun_out/r5_tails.txt; : > $OUT
for rep in 1 2; do for T in split serial; do
  echo "---- ULTRAGROTH_TAILS=$T: rank 5 of 8 at 2^24" >> $OUT
  ULTRAGROTH_TAILS=$T timeout -k 10 600 python3 tools/phase_times.py 24 8 5 9 products_first 8 U 2>/dev/null | grep "queued form" >> $OUT; tail -1 $OUT | cut -c1-250
done; done
for T in split serial; do
  echo "---- ULTRAGROTH_TAILS=$T: rank 0 of 8 at 2^24 (chain first)" >> $OUT
  ULTRAGROTH_TAILS=$T timeout -k 10 600 python3 tools/phase_times.py 24 8 0 9 chain_first 8 U 2>/dev/null | grep "queued form" >> $OUT; tail -1 $OUT | cut -c1-250
done
for rep in 1 2; do for T in split serial; do
  ULTRAGROTH_TAILS=$T timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --check > gpurun_out/r5_tails_b24_${T}_$rep.json 2>/dev/null
  ULTRAGROTH_TAILS=$T timeout -k 10 300 python3 bench.py --log-domain 20 --steps 20 --warmup 2 --bare > gpurun_out/r5_tails_b20_${T}_$rep.json 2>/dev/null
done; done
for T in split serial; do ULTRAGROTH_TAILS=$T timeout -k 10 300 python3 bench.py --ultra --log-domain 22 --steps 8 --warmup 2 --host-threads 1 --check > gpurun_out/r5_tails_ultra_$T.json 2>/dev/null; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_tails_b2*.json') + glob.glob('gpurun_out/r5_tails_ultra*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-34s ms/step %8.3f  unoverlapped %s  msm %.2f fft %.2f  %s" % (f.split('/')[-1], d["ms_per_step"], d.get("unoverlapped_ms_per_step"), d["msm_ms_per_proof"], d["fft_ms_per_proof"], d.get("check") or d["config"]["workload"][-22:]))
    except Exception as e:
        print(f, "FAILED", e)
PY

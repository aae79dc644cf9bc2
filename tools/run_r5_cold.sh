# Round 5: cold start -- create without waiting for the window tables; tests, then the figures at 2^24 (and ULTRAGROTH_TRACE of the first calls)
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5_cold_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5_cold_tests.log
for BG in 1 0; do
ULTRAGROTH_TABLES_BG=$BG ULTRAGROTH_TRACE=1 timeout -k 10 600 python3 bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/r5_cold_bench_bg$BG.json 2> gpurun_out/r5_cold_bench_bg$BG.err; echo "bench bg=$BG rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r5_cold_bench_bg$BG.json').read().strip().splitlines()[-1]); print('bg=$BG create %.3f first proof %.3f tables %.3f  ms/step %.2f api %.2f' % (d['create_s'], d['time_to_first_proof_s'], d['tables_in_use_after_s'], d['ms_per_step'], d['api_ms_per_step']))"
grep -m 12 "groth16" gpurun_out/r5_cold_bench_bg$BG.err
done

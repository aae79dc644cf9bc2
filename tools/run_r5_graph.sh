# Round 5: the recorded launch sequence (ULTRAGROTH_GRAPH=1) against eager launches, per-kernel event pairs on request only, the host
# part with windowed products: tests first, then A/B bench lines at 2^20 (configs[1] and full) and the default 2^24 line.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
ls /sys/class/drm/ > $O/r5_sysfs.txt 2>&1; for f in /sys/class/drm/card*/device/hwmon/hwmon*/; do echo "$f: $(ls $f | tr '\n' ' ')" >> $O/r5_sysfs.txt; done
timeout -k 10 900 python3 -m pytest tests/test_gpu_graph.py tests/test_gpu_faults.py -x -q -m gpu > $O/r5_graph_tests.log 2>&1; echo "graph tests rc=$?"; tail -3 $O/r5_graph_tests.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "bench_line_contract_one_gpu" > $O/r5_contract_test.log 2>&1; echo "contract test rc=$?"; tail -3 $O/r5_contract_test.log
for G in 0 1; do
  ULTRAGROTH_GRAPH=$G timeout -k 10 300 python3 bench.py --log-domain 20 --g1-only --steps 20 --warmup 2 --bare --overlap 0 > $O/r5_cfg1_g$G.json 2> $O/r5_cfg1_g$G.err; echo "cfg1 graph=$G rc=$?"
  ULTRAGROTH_GRAPH=$G timeout -k 10 300 python3 bench.py --log-domain 20 --steps 20 --warmup 2 --bare --overlap 0 > $O/r5_q20_g$G.json 2> $O/r5_q20_g$G.err; echo "2^20 graph=$G rc=$?"
  ULTRAGROTH_GRAPH=$G timeout -k 10 300 python3 bench.py --log-domain 20 --steps 20 --warmup 2 --bare --overlap 1 > $O/r5_q20_ov1_g$G.json 2> $O/r5_q20_ov1_g$G.err; echo "2^20 overlap graph=$G rc=$?"
done
for OV in 0 1 2; do
  timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --overlap $OV > $O/r5_b24_ov$OV.json 2> $O/r5_b24_ov$OV.err; echo "2^24 overlap=$OV rc=$?"
done
ULTRAGROTH_GRAPH=1 timeout -k 10 600 python3 bench.py --steps 8 --warmup 1 --bare --overlap 1 > $O/r5_b24_ov1_g1.json 2> $O/r5_b24_ov1_g1.err; echo "2^24 overlap=1 graph rc=$?"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_cfg1_g*.json') + glob.glob('gpurun_out/r5_q20*.json') + glob.glob('gpurun_out/r5_b24_ov*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-34s ms/step %8.3f  unoverlapped %s  msm %.2f fft %.2f  graph=%s overlap=%s" % (f.split('/')[-1], d["ms_per_step"], d.get("unoverlapped_ms_per_step"), d["msm_ms_per_proof"], d["fft_ms_per_proof"], d["config"].get("graph"), d["config"].get("overlap")))
    except Exception as e:
        print(f, "FAILED", e)
PY

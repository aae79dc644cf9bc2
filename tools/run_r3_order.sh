# round 3: where a chain rank of an 8-way shard starts its chain, and whether the stream priority class matters
# (tools/phase_times.py, rank 0 of 8 at 2^24): bash tools/run_r3_order.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/order.txt
for v in "h products_first" "n products_first" "h chain_thread" "n chain_thread" "h chain_first"; do
  set -- $v
  ULTRAGROTH_H_PRIORITY=$1 timeout -k 10 200 python3 tools/phase_times.py 24 8 0 9 $2 2>/dev/null | tail -n 2 >> gpurun_out/order.txt || exit 1
done
cat gpurun_out/order.txt

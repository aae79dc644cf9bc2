# Round 5, second batch of rank measurements (one rank of an N-way shard on one GPU, queued form):
#  * the N = 8 ranks at 2^24 and 2^26 again with this round's code (host part with windowed products, no per-kernel event pairs),
#  * UltraGroth 2^22 (configs[4]) for N = 2, 4, 8,
#  * a plain rank of eight at 2^24 with the window width and the reduction's lane counts tuned ON THE SHARD (UG_TABLE_C, UG_REDUCE_*_LOG).
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_rank_phases2.txt
: > $OUT
run() {   # log world rank wait_ms order  [env...]
  echo "---- 2^$1, rank $3 of $2 (wait $4 ms, $5) ${@:6}" >> $OUT
  env "${@:6}" timeout -k 10 600 python3 tools/phase_times.py $1 $2 $3 $4 $5 $2 U 2>> gpurun_out/r5_rank_phases2.err | grep -v amdgpu.ids >> $OUT || { echo "FAILED" >> $OUT; return 1; }
  tail -1 $OUT
}
run 24 8 5 9 products_first X=0 && run 24 8 0 9 chain_first X=0 && run 26 8 5 30 products_first X=0 && run 26 8 0 30 chain_first X=0 || exit 1
for W in 2 4 8; do
  for R in 0 $((W-1)); do
    echo "---- ultragroth 2^22, rank $R of $W" >> $OUT
    timeout -k 10 300 python3 tools/phase_times_ultra.py 22 $W $R 3 2>> gpurun_out/r5_rank_phases2.err | grep -v amdgpu.ids >> $OUT || { echo "FAILED" >> $OUT; exit 1; }
    tail -1 $OUT
  done
done
for C in 19 21; do run 24 8 5 9 products_first UG_TABLE_C=$C || exit 1; done
for G in 16 18; do run 24 8 5 9 products_first UG_REDUCE_G1_LOG=$G || exit 1; done
for G in 15 17; do run 24 8 5 9 products_first UG_REDUCE_G2_LOG=$G || exit 1; done
run 24 8 5 9 products_first UG_SEG_LANES_LOG=19 && run 24 8 5 9 products_first UG_SEG_LANES_LOG=21

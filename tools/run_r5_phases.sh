# Round 5: what ONE rank of an N-way shard costs for N = 2 and N = 4 (every distinct kind of rank: chain-heavy, chain, plain), queued
# form, at 2^24 and 2^26 -- the missing columns of DESIGN section 7's table (N = 8: profiles/r04_rank_phases_base.txt).
# bash tools/run_r5_phases.sh [sizes, default "24 26"]
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_rank_phases.txt
: > $OUT
run() {   # log world rank wait_ms order
  echo "---- 2^$1, rank $3 of $2 (wait $4 ms, $5)" >> $OUT
  timeout -k 10 900 python3 tools/phase_times.py $1 $2 $3 $4 $5 $2 U >> $OUT 2>> gpurun_out/r5_rank_phases.err || { echo "FAILED rc=$?" >> $OUT; return 1; }
  tail -1 $OUT
}
for L in ${1:-24 26}; do
  if [ $L = 24 ]; then W2=0; W4=8; else W2=0; W4=30; fi
  # two ranks: rank 0 runs chains 0 and 2, rank 1 chain 1 -- both are chain ranks, nobody waits for long
  run $L 2 0 $W2 products_first && run $L 2 1 $W2 products_first &&
  # four ranks: ranks 0..2 one chain each, rank 3 plain (waits for the chains' vectors)
  run $L 4 0 $W4 products_first && run $L 4 2 $W4 products_first && run $L 4 3 $W4 products_first || exit 1
done

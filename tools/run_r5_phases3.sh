# Round 5, third batch: the shard-tuned reduction (automatic now) on a plain and a chain rank of eight; a rank of THREE at 2^24 under
# rocprofv3 (what the kernels of one product family cost with n/3 points in flight: the paper costing of a section-major layout)
set -o pipefail
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_rank_phases3.txt
: > $OUT
for R in 5 0; do
  ORDER=products_first; [ $R = 0 ] && ORDER=chain_first
  echo "---- 2^24, rank $R of 8" >> $OUT
  timeout -k 10 600 python3 tools/phase_times.py 24 8 $R 9 $ORDER 8 U 2>/dev/null | grep -v "amdgpu\|^order" >> $OUT; tail -1 $OUT
done
echo "---- 2^24, rank 1 of 3 under rocprofv3" >> $OUT
bash tools/run_prof_rank.sh 24 3 1 3 U >> $OUT 2>&1
tail -45 $OUT

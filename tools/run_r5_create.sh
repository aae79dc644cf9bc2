# where groth16_prover_create spends its time, three processes in a row (the first step varies between 0.08 and 1.1 s)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  ULTRAGROTH_TRACE=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --bare 2> gpurun_out/r5_create_$i.err > /dev/null
  echo "== process $i"; grep -m 12 "ug_api\|create:" gpurun_out/r5_create_$i.err
done

# what could a radix-8 / radix-16 NTT step buy? The -DUG_MEASURE library with UG_NTT_FUSE_STEPS=1 runs two
# radix-4 steps per LDS round trip (WRONG results: timing only), an upper bound for any higher radix. A/B on one box.
cd $GRAFT_REPO_ROOT
export ULTRAGROTH_LIB=$GRAFT_REPO_ROOT/ultragroth_amd/csrc/libultragroth_hip_measure.so
for F in 0 1 0 1; do
UG_NTT_FUSE_STEPS=$F python bench.py --log-domain 24 --steps 5 --warmup 1 --no-cpu-baseline --host-threads 1 > gpurun_out/ntt_fuse_$F.json 2> gpurun_out/ntt_fuse_$F.err
python - <<PY
import json
d = json.loads(open("gpurun_out/ntt_fuse_$F.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]["ntt_pass_kernel"]
print("UG_NTT_FUSE_STEPS=$F  ntt_pass_kernel avg launch %.4f ms x %d launches per 5 steps = %.3f ms per proof; fft_ms_per_proof %.3f; ms_per_step %.2f" % (k["avg_launch_ms"], k["launches"], k["ms_per_step"], d["fft_ms_per_proof"], d["ms_per_step"]))
PY
done

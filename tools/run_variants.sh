# A/B of kernel variants built as whole libraries under ultragroth_amd/csrc/build/variants/ (tools/README.md):
#   bash tools/run_variants.sh v1 v2 ...      each: 2^24 bench with --check (bit-exact or exit), per-kernel split
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp ultragroth_amd/csrc/libultragroth_hip.so /tmp/libug_base.so
for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/libug_base.so ultragroth_amd/csrc/libultragroth_hip.so
    else cp ultragroth_amd/csrc/build/variants/libug_$v.so ultragroth_amd/csrc/libultragroth_hip.so; fi
    echo "== $v" | tee -a gpurun_out/variants.log
    bash tools/run_b24.sh --check --host-threads 1 2>&1 | tee -a gpurun_out/variants.log || exit 1
done
cp /tmp/libug_base.so ultragroth_amd/csrc/libultragroth_hip.so

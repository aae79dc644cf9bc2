#!/bin/bash
# Round 4 (VERDICT r3 "small" item): the reference's LARGEST legal domain, 2^27 (src/groth16.hpp:109: FFT(domainSize * 2), Fr has
# 2-adicity 28), through the piecewise path at its own size: the full-size whole-proof test and the 8-rank rehearsal with
# UG_HUGE_LOG=27 (75 GB zkey in host memory, ranges above 2^26 scalars proved in pieces), then the single-GPU figure at 2^26.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2
UG_HUGE_LOG=27 timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "configs3_size" --durations=5 > gpurun_out/r4_2p27.log 2>&1; echo "2^27 rc=$?"
tail -12 gpurun_out/r4_2p27.log
grep "2^27" gpurun_out/fullsize_progress.log | tail -30

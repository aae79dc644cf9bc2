#!/bin/bash
# round 3: the whole -m gpu suite as the driver runs it, then smoke(), then the driver's default bench line
set -o pipefail
mkdir -p gpurun_out
( time timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/r3_full_suite.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r3_full_suite.log; tail -25 gpurun_out/r3_full_suite.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

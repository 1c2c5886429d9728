#!/bin/bash
# round 6, GPU call 45: where the -m gpu suite spends its time (--durations=40), on the final tree
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_45
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
nproc; cat /proc/loadavg
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=40 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | grep "passed\|failed\|FAILED\|Error" | tail -3
grep -A45 "slowest 40 durations" $OUT/gpu_suite.log | head -48 | cut -c1-200
cat /proc/loadavg

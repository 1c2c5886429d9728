#!/bin/bash
# round 6, GPU call 25: the warm append again, with the duration of the helper thread's half of the build
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_25
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py --warmup > $OUT/shim.txt 2>&1
grep -v "Ceres Solver Report" $OUT/shim.txt | tail -48

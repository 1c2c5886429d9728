#!/bin/bash
# round 6, GPU call 24: the warm adjustBundle() with one added view, split further (append: check / slots / build / final sync)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_24
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py --warmup > $OUT/shim.txt 2>&1
grep -v "Ceres Solver Report" $OUT/shim.txt | tail -48

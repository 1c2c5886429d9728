#!/bin/bash
# round 6, GPU call 26: build_structure -- the parameter upload on the helper thread, the solve's buffers carved before the join: append / shim / sharded / edge tests, then the shim timing
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_26
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 600 -x -k "append or shim or edge or sharded or capi or determin or matrix_free or baseline" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; grep -v "Ceres Solver Report" $OUT/tests.log | tail -5
cd /tmp
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py --warmup > $OUT/shim_marks.txt 2>&1
grep "helper\|alloc buffers\|join\|descriptors\|adjustBundle() wall\|path:\|build_structure" $OUT/shim_marks.txt | tail -14
python $REPO/tools/time_shim_incremental.py --warmup > $OUT/shim.txt 2>&1
grep "adjustBundle() wall\|path:\|marshal " $OUT/shim.txt | tail -12

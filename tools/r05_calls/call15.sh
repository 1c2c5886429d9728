#!/bin/bash
# round 5, GPU call 15: F32J accuracy against the oracle, this tree against the previous commit (tools/ab/head)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_15; mkdir -p $OUT; cd $REPO
echo "== new ==" > $OUT/f32j_accuracy.txt; python tools/f32j_accuracy.py 2>/dev/null | grep -v "Ceres Solver" >> $OUT/f32j_accuracy.txt
echo "== previous commit ==" >> $OUT/f32j_accuracy.txt; SFMBA_LIB=$REPO/tools/ab/head/libsfmba_hip.so python tools/f32j_accuracy.py 2>/dev/null | grep -v "Ceres Solver" >> $OUT/f32j_accuracy.txt
cat $OUT/f32j_accuracy.txt

#!/bin/bash
# round 6, GPU call 10: camera counts the suite had never seen: 2 500 cameras (d = 15 001) and 7 800 cameras (d = 46 801: d * ld > 2^31)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_10
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 600 python $REPO/tools/large_cameras_check.py 2500 > $OUT/large_2500.txt 2>&1; echo "rc=$?" >> $OUT/large_2500.txt; cat $OUT/large_2500.txt | grep -v "Ceres Solver"
timeout -k 5 900 python $REPO/tools/large_cameras_check.py 7800 30000 f32j_pcg,f64_pcg > $OUT/large_7800.txt 2>&1; echo "rc=$?" >> $OUT/large_7800.txt; cat $OUT/large_7800.txt | grep -v "Ceres Solver"

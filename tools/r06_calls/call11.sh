#!/bin/bash
# round 6, GPU call 11: 7 800 cameras (d * ld > 2^31), three LM iterations, the factorisation included
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_11
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 5 900 python $REPO/tools/large_cameras_check.py 7800 30000 f32j_pcg,f64_pcg,f64_chol 3 > $OUT/large_7800_3its.txt 2>&1; echo "rc=$?" >> $OUT/large_7800_3its.txt; cat $OUT/large_7800_3its.txt | grep -v "Ceres Solver"

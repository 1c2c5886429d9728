#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05_8
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_matrix_free.py tests/test_gpu_segments.py -x -q --timeout 300 -k "row_two_cams or row_tiny or small_fixtures or other_hat_counts" > $OUT/new.log 2>&1
echo "new rc=$?" >> $OUT/new.log; grep -v "Ceres Solver Report" $OUT/new.log | tail -30
timeout -k 5 1200 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | tail -6
python bench.py --sharded-extras 1 --steps 5 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | cut -c1-100

#!/bin/bash
# round 6, GPU call 39: tests/test_gpu_edge_cases.py again, with a handle that is given non-finite parameters and finite ones in turn
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_39
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --timeout 600 --durations=5 > $OUT/edge.log 2>&1; grep -v "Ceres Solver Report" $OUT/edge.log | tail -14 | cut -c1-300

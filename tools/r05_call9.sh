#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05_9
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -x -q --timeout 400 --durations=15 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | tail -28
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 | cut -c1-160

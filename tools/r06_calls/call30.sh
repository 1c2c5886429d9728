#!/bin/bash
# round 6, GPU call 30: case 103 of the fuzz sequence (seed 21) alone: a resident F32J handle that ends in FAILURE and is not repeatable after reset
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_30
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 600 python tests/fuzz_parity.py --cases 104 --seed 21 --only 103 > $OUT/case103.txt 2>&1
grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/case103.txt | tail -70
timeout -k 5 900 python tests/fuzz_parity.py --cases 400 --seed 21 > $OUT/fuzz_small_after.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/fuzz_small_after.txt | tail -30
timeout -k 5 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "cholesky or dense or edge or baseline_parity or sharded or cfg2" > $OUT/tests.log 2>&1; grep -v "Ceres Solver Report" $OUT/tests.log | tail -3

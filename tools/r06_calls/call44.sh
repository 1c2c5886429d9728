#!/bin/bash
# round 6, GPU call 44: the iteration-limit test again (the oracle now checks the iteration count before the gradient tolerance, as Ceres does); options sweep
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_44
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --timeout 600 --durations=5 > $OUT/edge.log 2>&1; grep -v "Ceres Solver Report" $OUT/edge.log | tail -14 | cut -c1-300
timeout -k 5 900 python tests/fuzz_parity.py --cases 400 --seed 81 --options > $OUT/opt_81.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/opt_81.txt | grep -v "^        \|inexact mode\| soft:" | tail -16 | cut -c1-300

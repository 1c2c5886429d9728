#!/bin/bash
# round 6, GPU call 31: the regression test of the poisoned padding rows; two more fuzz sequences (single-threaded oracle)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_31
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --timeout 300 > $OUT/edge.log 2>&1; grep "passed\|failed\|Error\|assert" $OUT/edge.log | tail -5
timeout -k 5 1500 python tests/fuzz_parity.py --cases 600 --seed 31 > $OUT/fuzz_31.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/fuzz_31.txt | tail -40
timeout -k 5 1500 python tests/fuzz_parity.py --cases 80 --seed 32 --big > $OUT/fuzz_32.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/fuzz_32.txt | tail -20

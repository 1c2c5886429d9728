#!/bin/bash
# round 6, GPU call 34: the edge-case tests with the fuzz sweep as a test (tests/fuzz_parity.py --cases 250 --seed 21)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_34
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q --timeout 600 > $OUT/edge.log 2>&1; grep -v "Ceres Solver Report" $OUT/edge.log | tail -12 | cut -c1-300
python tests/fuzz_parity.py --cases 250 --seed 21 2>&1 | grep -v "Ceres Solver Report\|amdgpu.ids" | grep "HARD\|long run\|fuzz_parity:" | cut -c1-400

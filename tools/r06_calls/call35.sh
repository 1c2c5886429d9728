#!/bin/bash
# round 6, GPU call 35: tests/fuzz_handles.py -- growing handles (append), deterministic handles, matrix-free handles, set_params: 400 random cases against the oracle
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_35
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
for sd in 51 52; do timeout -k 5 1500 python tests/fuzz_handles.py --cases 200 --seed $sd > $OUT/handles_$sd.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/handles_$sd.txt | tail -25 | cut -c1-330; done

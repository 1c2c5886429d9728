#!/bin/bash
# round 6, GPU call 36: tests/fuzz_sharded.py -- the four sharded forms on one rank against the oracle, 300 random cases
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_36
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
for sd in 61 62; do timeout -k 5 1500 python tests/fuzz_sharded.py --cases 150 --seed $sd > $OUT/sharded_$sd.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/sharded_$sd.txt | tail -25 | cut -c1-360; done

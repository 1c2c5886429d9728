#!/bin/bash
# round 6, GPU call 29: randomised parity sweep against the oracle (tests/fuzz_parity.py): 300 small / medium cases, then 40 with shapes above 213 cameras
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_29
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 1500 python tests/fuzz_parity.py --cases 400 --seed 21 > $OUT/fuzz_small.txt 2>&1; echo "rc=$?" >> $OUT/fuzz_small.txt
grep -v "Ceres Solver Report" $OUT/fuzz_small.txt | tail -25
timeout -k 5 1500 python tests/fuzz_parity.py --cases 60 --seed 22 --big > $OUT/fuzz_big.txt 2>&1; echo "rc=$?" >> $OUT/fuzz_big.txt
grep -v "Ceres Solver Report" $OUT/fuzz_big.txt | tail -15

#!/bin/bash
# round 6, GPU call 33: F32J divides the radius by 8 after an invalid step -- the whole -m gpu suite, then four fuzz sequences
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_33
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | grep "passed\|failed\|FAILED\|Error" | tail -5
for sd in 41 42 43; do timeout -k 5 1200 python tests/fuzz_parity.py --cases 500 --seed $sd > $OUT/fuzz_$sd.txt 2>&1; grep "HARD\|not repeatable\|FAILURE\|EXCEPTION\|fuzz_parity:" $OUT/fuzz_$sd.txt | cut -c1-330 | tail -12; done
timeout -k 5 1500 python tests/fuzz_parity.py --cases 100 --seed 44 --big > $OUT/fuzz_44_big.txt 2>&1; grep "HARD\|not repeatable\|FAILURE\|EXCEPTION\|fuzz_parity:" $OUT/fuzz_44_big.txt | cut -c1-330 | tail -12

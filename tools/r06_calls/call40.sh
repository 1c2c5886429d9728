#!/bin/bash
# round 6, GPU call 40: tests/fuzz_parity.py --options -- random LM options (iteration limits, radii, tolerances, Jacobi scaling off, diagonal clamp ...) on both sides
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_40
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
for sd in 81 82; do timeout -k 5 900 python tests/fuzz_parity.py --cases 400 --seed $sd --options > $OUT/opt_$sd.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/opt_$sd.txt | grep -v "^        \|inexact mode\| soft:" | tail -24 | cut -c1-420; done

#!/bin/bash
# round 6, GPU call 32: case 34 of the --big fuzz sequence (seed 32) alone: an F32J + Cholesky handle (230 cameras, banded, outliers) whose second solve differs from its first
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_32
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python tests/fuzz_parity.py --cases 35 --seed 32 --big --only 34 > $OUT/case34.txt 2>&1
grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/case34.txt | grep -v "^   it " | tail -40

#!/bin/bash
# round 6, GPU call 43: the sweeps on shapes above 213 cameras (streaming CG on one triangle): handle variants incl. append, and the parity sweep with random options
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_43
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 1200 python tests/fuzz_handles.py --cases 60 --seed 91 --big > $OUT/handles_big.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/handles_big.txt | tail -14 | cut -c1-330
timeout -k 5 1200 python tests/fuzz_parity.py --cases 80 --seed 92 --big --options > $OUT/parity_big_opt.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/parity_big_opt.txt | grep -v "^        \|inexact mode\| soft:" | tail -14 | cut -c1-330

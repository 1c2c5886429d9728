#!/bin/bash
# round 6, GPU call 47: tools/leak_check.py -- 400 rounds of create / solve / reset / solve / destroy over five shapes, both precisions, all handle kinds: RSS and free device memory
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_47
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python tools/leak_check.py 400 > $OUT/leak.txt 2>&1; grep -v "Ceres Solver Report\|amdgpu.ids" $OUT/leak.txt | tail -24

#!/bin/bash
# round 6, GPU call 46: tests/fuzz_association.py -- find2D3DMatches / mergeNewPointCloud over 200 more seeds and sizes, exact comparison with the oracle
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_46
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 1500 python tests/fuzz_association.py --seeds 200 --first 1000 > $OUT/assoc.txt 2>&1; tail -20 $OUT/assoc.txt | cut -c1-300

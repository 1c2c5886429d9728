#!/bin/bash
# round 5, GPU call 6: sharded tests after the dcg kernel changes, then the evidence sets r05_a (cfg 3) and r05_a (cfg 5)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd $REPO
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_6
timeout -k 5 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_options_v4.py tests/test_gpu_matrix_free.py -x -q --timeout 300 > gpurun_out/r05_6/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_6/tests.log; tail -4 gpurun_out/r05_6/tests.log
timeout 1500 bash tools/collect_evidence.sh r05_a > gpurun_out/r05_6/collect3.log 2>&1; tail -3 gpurun_out/r05_6/collect3.log
timeout 600 bash tools/collect_cfg5.sh r05_a > gpurun_out/r05_6/collect5.log 2>&1; tail -3 gpurun_out/r05_6/collect5.log

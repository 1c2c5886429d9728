#!/bin/bash
# round 6, GPU call 13: the evidence set r06_e (tools/collect_evidence.sh: bench lines, kernel statistics, counter traffic, sharded forms on one rank, the shim)
# and the BASELINE config 5 set (tools/collect_cfg5.sh)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
cd $REPO
bash tools/collect_evidence.sh r06_e > gpurun_out/collect_r06_e.log 2>&1
bash tools/collect_cfg5.sh r06_e > gpurun_out/collect5_r06_e.log 2>&1
ls gpurun_out/evidence gpurun_out/evidence5 | head -60

#!/bin/bash
# round 6, GPU call 27: the evidence set of the final tree (tools/collect_evidence.sh r06_f, tools/collect_cfg5.sh r06_f)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
bash $REPO/tools/collect_evidence.sh r06_f > $REPO/gpurun_out/collect_r06_f.log 2>&1
bash $REPO/tools/collect_cfg5.sh r06_f > $REPO/gpurun_out/collect5_r06_f.log 2>&1
tail -3 $REPO/gpurun_out/collect_r06_f.log; tail -3 $REPO/gpurun_out/collect5_r06_f.log

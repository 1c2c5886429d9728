#!/bin/bash
# round 5, GPU call 5: fused control (two-level tickets) against the control launch of its own, cfg 3 and cfg 5
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_5
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for r in 1 2 3; do for v in cur nofuse; do
  LIB=$REPO/sfm-toy-library_amd/csrc/libsfmba_hip.so
  [ $v != cur ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  SFMBA_LIB=$LIB timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('$v cfg3', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'point_update %.1f control %.1f' % (k.get('point_update',0), k.get('control',0)), 'auto', round(d['default_solver_auto']['value'],1))" >> $OUT/ab_fuse.txt 2>&1
done; done
for v in cur nofuse; do
  LIB=$REPO/sfm-toy-library_amd/csrc/libsfmba_hip.so
  [ $v != cur ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  SFMBA_LIB=$LIB timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('$v cfg5', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'point_update %.1f control %.1f' % (k.get('point_update',0), k.get('control',0)))" >> $OUT/ab_fuse.txt 2>&1
done
cat $OUT/ab_fuse.txt
timeout 300 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_rejections.py tests/test_gpu_fullsize.py -x -q --timeout 300 2>&1 | tail -3

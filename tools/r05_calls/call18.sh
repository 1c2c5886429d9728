#!/bin/bash
# round 5, GPU call 18: the trial sweep of the F32J back-substitution with the trial pose rebuilt from the six camera parameters (three gathers + sincos per observation; default)
# against the stored [R | t] (six gathers; tools/ab/tabpose): bench cfg 3 / cfg 5, two rounds; then the parity suites that see the trial cost
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_18
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for rep in 1 2; do
for wl in cfg3 cfg5; do
for v in default tabpose; do
  LIB=""; [ $v != default ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  ST=5; [ $wl = cfg3 ] && ST=40
  SFMBA_LIB=$LIB python bench.py --workload $wl --steps $ST --warmup 3 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | grep '^{' > $OUT/${wl}_$v.json
  python - <<PY
import json
d=json.loads(open("$OUT/${wl}_$v.json").read())
ks={k["kernel"]: k.get("avg_launch_us") for k in d.get("roofline_all_kernels", []) if "kernel" in k}
print("%-6s %-10s %8.1f it/s   point_update %s us (event brackets)" % ("$wl", "$v", d["value"], round(ks.get("point_update") or 0, 1)))
PY
done; done; done 2>&1 | tee $OUT/ab_trial_pose.txt
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_rejections.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -q --timeout 400 2>&1 | grep "passed\|failed\|FAILED" | tail -5

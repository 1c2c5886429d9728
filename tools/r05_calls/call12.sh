#!/bin/bash
# round 5, GPU call 12: pair pass -- two-stage epilogue (default) against the one-stage form (tools/ab/oldepi), and the sixteen-lane form additionally without the
# point-table look-ahead (tools/ab/noprefetch = the round-4 kernel); cfg 3 and cfg 5, two rounds; then the parity tests of the pair forms
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_12
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for rep in 1 2; do
for wl in cfg3 cfg5; do
for v in default oldepi noprefetch; do
  LIB=""; [ $v != default ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  ST=5; [ $wl = cfg3 ] && ST=40
  SFMBA_LIB=$LIB python bench.py --workload $wl --steps $ST --warmup 3 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | grep '^{' > $OUT/${wl}_$v.json
  python - <<PY
import json
d=json.loads(open("$OUT/${wl}_$v.json").read())
ks={k["kernel"]: k.get("avg_launch_us") for k in d.get("roofline_all_kernels", []) if "kernel" in k}
print("%-6s %-10s %8.1f it/s   pair pass %s us (event brackets)" % ("$wl", "$v", d["value"], round(ks.get("schur_pairs") or 0, 1)))
PY
done; done; done 2>&1 | tee $OUT/ab_pair_epilogue.txt
timeout -k 5 900 python -m pytest tests/test_gpu_pair_forms.py tests/test_gpu_baseline_parity.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q --timeout 400 2>&1 | grep -v "Ceres Solver Report" | tail -4

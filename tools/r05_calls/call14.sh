#!/bin/bash
# round 5, GPU call 14: F32J back-substitution with the fp32 camera records in its first sweep: bench cfg 3 / cfg 5 (two rounds), kernel stats, then the whole GPU suite
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_14
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for rep in 1 2; do
for wl in cfg3 cfg5; do
  ST=5; [ $wl = cfg3 ] && ST=40
  python bench.py --workload $wl --steps $ST --warmup 3 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | grep '^{' > $OUT/${wl}.json
  python - <<PY
import json
d=json.loads(open("$OUT/${wl}.json").read())
ks={k["kernel"]: k.get("avg_launch_us") for k in d.get("roofline_all_kernels", []) if "kernel" in k}
print("%-6s %8.1f it/s   point_update %s us (event brackets)  final cost %r" % ("$wl", d["value"], round(ks.get("point_update") or 0, 1), d.get("config", {}).get("final_cost")))
PY
done; done 2>&1 | tee $OUT/bench.txt
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | tail -25

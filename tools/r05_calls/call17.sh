#!/bin/bash
# round 5, GPU call 17: the fp32 camera records in the sharded loops as well: whole GPU suite, then the one-rank sharded bench lines (r05_b)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_17
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | grep "passed\|failed\|FAILED\|rc=" | tail -8
cd /tmp
BENCH="python $REPO/bench.py"
for wl in cfg3 cfg5; do
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic 2> /dev/null | grep '^{' > $OUT/r05_b_${wl}_sharded_1rank_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --distributed-cg 2> /dev/null | grep '^{' > $OUT/r05_b_${wl}_sharded_1rank_distributed_cg_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --implicit-cg 2> /dev/null | grep '^{' > $OUT/r05_b_${wl}_sharded_1rank_implicit_cg_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --row-sharded 2> /dev/null | grep '^{' > $OUT/r05_b_${wl}_sharded_1rank_row_sharded_bench.json
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r05_b_*bench.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); nc=(d.get("config",{}) or {}).get("one_rank_without_collective") or {}
    print("%-62s %8.1f   without a communicator call: %s" % (f.split('/')[-1], d["value"], round(nc.get("value",0),1)))
PY

#!/bin/bash
# round 5, GPU call 19: trial pose from the camera parameters at >= 400 cameras: whole GPU suite, the cfg 5 evidence set again (r05_b), the one-rank sharded lines at cfg 5, the default bench line
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_19
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 600 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | grep "passed\|failed\|FAILED\|rc=" | tail -8
bash tools/collect_cfg5.sh r05_b > $OUT/collect5.log 2>&1
cd /tmp
BENCH="python $REPO/bench.py"
$BENCH --mode sharded --workload cfg5 --steps 5 --no-cpu-baseline --no-live-traffic 2> /dev/null | grep '^{' > $OUT/r05_b_cfg5_sharded_1rank_bench.json
$BENCH --mode sharded --workload cfg5 --steps 5 --no-cpu-baseline --no-live-traffic --distributed-cg 2> /dev/null | grep '^{' > $OUT/r05_b_cfg5_sharded_1rank_distributed_cg_bench.json
$BENCH --mode sharded --workload cfg5 --steps 5 --no-cpu-baseline --no-live-traffic --implicit-cg 2> /dev/null | grep '^{' > $OUT/r05_b_cfg5_sharded_1rank_implicit_cg_bench.json
$BENCH --mode sharded --workload cfg5 --steps 5 --no-cpu-baseline --no-live-traffic --row-sharded 2> /dev/null | grep '^{' > $OUT/r05_b_cfg5_sharded_1rank_row_sharded_bench.json
rm -rf $OUT/stats_row
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_row -- $BENCH --mode sharded --workload cfg5 --row-sharded --steps 4 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_row.err
python $REPO/tools/rocprof_summary.py $OUT/stats_row $OUT/r05_b_cfg5_sharded_1rank_row_sharded_kernel_stats.txt "r05_b: bench.py --mode sharded --workload cfg5 --row-sharded --steps 4 (one rank, RCCL communicator of one rank) under rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $OUT/stats_row
$BENCH --steps 20 --warmup 5 2> /dev/null | tail -1 > $OUT/r05_b_cfg3_pcg_bench.json
python - <<PY
import json, glob
def find(d, key):
    if isinstance(d, dict):
        if key in d: return d[key]
        for v in d.values():
            r=find(v,key)
            if r is not None: return r
    return None
for f in sorted(glob.glob("$OUT/r05_b_*bench.json")) + sorted(glob.glob("$REPO/gpurun_out/evidence5/r05_b_*bench.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); nc=find(d,"one_rank_without_collective") or {}
    print("%-62s %8.1f  ms/step %.4f  no comm: %s" % (f.split('/')[-1], d["value"], d["ms_per_step"], round(nc.get("value",0),1) if isinstance(nc,dict) else nc))
PY

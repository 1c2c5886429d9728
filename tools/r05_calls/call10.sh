#!/bin/bash
# counter traffic of the row-sharded solve's kernels (cfg 5, one rank), a 200-step headline run, the whole GPU suite
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH --mode sharded --workload cfg5 --row-sharded --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/pmc_$c.err
done
python $REPO/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/r05_b_cfg5_sharded_1rank_row_sharded_pmc_traffic.txt "python bench.py --mode sharded --workload cfg5 --row-sharded --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic   (cfg5, f32j, one rank)" | head -30
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
$BENCH --steps 200 --warmup 5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | grep '^{' > $OUT/r05_b_cfg3_pcg_bench_200steps.json
python - <<PY
import json
d=json.loads(open("$OUT/r05_b_cfg3_pcg_bench_200steps.json").read().strip().splitlines()[-1]); print("200 steps:", d["value"], d["ms_per_step"])
PY
cd $REPO
timeout -k 5 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | tail -8

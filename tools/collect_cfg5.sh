#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the BASELINE config 5 evidence set (1000 cams / 500k pts / 5M obs, one GPU) -> gpurun_out/evidence5/ ;
# copy what is to be judged into profiles/.      tools/collect_cfg5.sh <tag> [extra bench args]
set -u
TAG=${1:-rXX}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/evidence5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic $*"
# 1. the bench line (unsharded, one resident problem) and the sharded loop on this box's one rank, both CG forms
$BENCH --steps 5 --warmup 2 > $OUT/${TAG}_cfg5_pcg_bench.json 2> $OUT/bench5.err
# 2. rocprofv3 kernel statistics of the same command
rm -rf $OUT/stats5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- $BENCH --steps 4 --warmup 1 > /dev/null 2> $OUT/stats5.err
python $REPO/tools/rocprof_summary.py $OUT/stats5 $OUT/${TAG}_cfg5_pcg_kernel_stats.txt "$TAG: bench.py --workload cfg5 --steps 4 --warmup 1 (f32j, PCG) under rocprofv3 --kernel-trace --stats" > /dev/null
# 3. HBM-side traffic: separate PMC passes, counters only
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc5_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc5_$c -- $BENCH --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc5_$c.err
done
python $REPO/tools/pmc_summary.py $OUT/pmc5_FETCH_SIZE $OUT/pmc5_WRITE_SIZE $OUT/${TAG}_cfg5_pcg_pmc_traffic.txt "python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic   (cfg5, f32j, PCG)" > /dev/null
rm -rf $OUT/stats5 $OUT/pmc5_FETCH_SIZE $OUT/pmc5_WRITE_SIZE
ls -la $OUT

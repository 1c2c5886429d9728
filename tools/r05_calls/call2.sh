#!/bin/bash
# round 5, GPU call 2: the multi-rank tests of the new paths, then one-rank kernel statistics of the row-sharded and the distributed solve at cfg 5
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_2
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_matrix_free.py tests/test_gpu_sharded.py -q --timeout 600 -k "matrix_free or pair_limit or no_pair or two_to_the or row_ or rccl_bindings or dist_cfg2 or dist_wide or dist_auto or dist_plain or dist_imp_cfg2 or duplicate" > $OUT/new_tests.log 2>&1
echo "new tests rc=$?" >> $OUT/new_tests.log
tail -40 $OUT/new_tests.log
cd /tmp
for v in row-sharded distributed-cg; do
  rm -rf $OUT/stats_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$v -- python $REPO/bench.py --mode sharded --workload cfg5 --$v --steps 4 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_$v.err
  python $REPO/tools/rocprof_summary.py $OUT/stats_$v $OUT/r05_a_cfg5_sharded_1rank_${v}_kernel_stats.txt "r05_a: bench.py --mode sharded --workload cfg5 --$v --steps 4 (one rank) under rocprofv3 --kernel-trace --stats" > /dev/null
  rm -rf $OUT/stats_$v
done
head -40 $OUT/r05_a_cfg5_sharded_1rank_row-sharded_kernel_stats.txt

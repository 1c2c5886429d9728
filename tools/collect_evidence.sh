#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the evidence set of a round -> gpurun_out/evidence/ ; copy what is to be judged into profiles/.
#   tools/collect_evidence.sh <tag>       e.g. r01_c
set -u
TAG=${1:-rXX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/evidence
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py"
# 1. the bench line itself (with cpu_baseline), PCG (default) and the exact Cholesky path
$BENCH > $OUT/${TAG}_cfg3_pcg_bench.json 2> $OUT/bench_pcg.err
# (the same default run carries `extra_workloads`: cfg 2 in fp64, cfg 3 in fp64, cfg3_banded, cfg 5 -- each held to the oracle's stored result)
$BENCH --linear cholesky --no-cpu-baseline --no-live-traffic --extra-workloads 0 > $OUT/${TAG}_cfg3_cholesky_bench.json 2> $OUT/bench_chol.err
# 2. rocprofv3 kernel statistics of the same command
for lin in pcg cholesky; do
  rm -rf $OUT/stats_$lin
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$lin -- $BENCH --linear $lin --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_$lin.err
  python $REPO/tools/rocprof_summary.py $OUT/stats_$lin $OUT/${TAG}_cfg3_${lin}_kernel_stats.txt "$TAG: bench.py --linear $lin --steps 10 --warmup 2 (cfg3, f32j) under rocprofv3 --kernel-trace --stats" > /dev/null
done
# 3. HBM-side traffic: separate PMC passes, no trace domains besides --kernel-trace
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/pmc_$c.err
done
python $REPO/tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/${TAG}_cfg3_pcg_pmc_traffic.txt "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic   (cfg3, f32j, PCG)" > /dev/null
# 3b. the library default (AUTO) and the realistic-visibility workload
$BENCH --linear auto --no-cpu-baseline --no-live-traffic --extra-workloads 0 > $OUT/${TAG}_cfg3_auto_bench.json 2> $OUT/bench_auto.err
$BENCH --workload cfg3_banded --no-cpu-baseline --no-live-traffic --extra-workloads 0 > $OUT/${TAG}_cfg3_banded_pcg_bench.json 2> $OUT/bench_banded.err
rm -rf $OUT/stats_banded
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_banded -- $BENCH --workload cfg3_banded --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_banded.err
python $REPO/tools/rocprof_summary.py $OUT/stats_banded $OUT/${TAG}_cfg3_banded_pcg_kernel_stats.txt "$TAG: bench.py --workload cfg3_banded --steps 10 --warmup 2 (f32j, PCG) under rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $OUT/stats_banded
$REPO/tools/micro/pk_bench > $OUT/${TAG}_valu_issue_microbench.txt 2>&1
# 4. the sharded path on this box's one rank (RCCL communicator of one rank; the exchange is a no-op, its pack / unpack kernels are not):
#    replicated CG, distributed CG (reduce-scatter of the blocks), implicit Schur CG (no exchange of the reduced matrix)
for wl in cfg3 cfg5; do
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic 2> $OUT/sharded_$wl.err | grep '^{' > $OUT/${TAG}_${wl}_sharded_1rank_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --distributed-cg 2> $OUT/sharded_d_$wl.err | grep '^{' > $OUT/${TAG}_${wl}_sharded_1rank_distributed_cg_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --implicit-cg 2> $OUT/sharded_i_$wl.err | grep '^{' > $OUT/${TAG}_${wl}_sharded_1rank_implicit_cg_bench.json
  $BENCH --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic --row-sharded 2> $OUT/sharded_r_$wl.err | grep '^{' > $OUT/${TAG}_${wl}_sharded_1rank_row_sharded_bench.json
done
# one-rank kernel statistics of the row-sharded solve at BASELINE config 5 (fourth column)
rm -rf $OUT/stats_row
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_row -- $BENCH --mode sharded --workload cfg5 --row-sharded --steps 4 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_row.err
python $REPO/tools/rocprof_summary.py $OUT/stats_row $OUT/${TAG}_cfg5_sharded_1rank_row_sharded_kernel_stats.txt "$TAG: bench.py --mode sharded --workload cfg5 --row-sharded --steps 4 (one rank, RCCL communicator of one rank) under rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $OUT/stats_row
rm -rf $OUT/stats_sh
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sh -- $BENCH --mode sharded --workload cfg3 --steps 20 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/stats_sh.err
python $REPO/tools/rocprof_summary.py $OUT/stats_sh $OUT/${TAG}_cfg3_sharded_1rank_kernel_stats.txt "$TAG: bench.py --mode sharded --workload cfg3 --steps 20 (one rank, RCCL communicator of one rank) under rocprofv3 --kernel-trace --stats" > /dev/null
rm -rf $OUT/stats_sh
# 5. the drop-in shim in the reference's call pattern: one view added to 199 (SfM.cpp:464-466)
echo "== SFMBA_LINEAR=pcg (block-Jacobi + gauge coarse space CG) ==" > $OUT/${TAG}_shim_incremental.txt
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py >> $OUT/${TAG}_shim_incremental.txt 2>&1
echo "== SFMBA_LINEAR=auto (the shim's default since round 3: DENSE_SCHUR result through the CG at 1e-12, Cholesky fallback) ==" >> $OUT/${TAG}_shim_incremental.txt
SFMBA_LINEAR=auto python $REPO/tools/time_shim_incremental.py >> $OUT/${TAG}_shim_incremental.txt 2>&1
echo "== SFMBA_LINEAR=cholesky (always factorise: the reference's literal configuration) ==" >> $OUT/${TAG}_shim_incremental.txt
SFMBA_LINEAR=cholesky python $REPO/tools/time_shim_incremental.py >> $OUT/${TAG}_shim_incremental.txt 2>&1
echo "== structure build on its own ==" >> $OUT/${TAG}_shim_incremental.txt
python $REPO/tools/time_create.py >> $OUT/${TAG}_shim_incremental.txt 2>&1
# keep the merge-back small
rm -rf $OUT/stats_pcg $OUT/stats_cholesky $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT

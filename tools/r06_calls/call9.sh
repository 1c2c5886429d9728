#!/bin/bash
# round 6, GPU call 9: the F32J error budget (measured values), sfmba_device_warmup, the roctx ranges under rocprofv3 --marker-trace, shim timing with and without the warm-up
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_9
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_f32j_budget.py tests/test_gpu_edge_cases.py -m gpu -q -s --timeout 600 > $OUT/budget.log 2>&1
echo "rc=$?" >> $OUT/budget.log; grep "it (\|passed\|failed\|rc=\|Error" $OUT/budget.log | tail -48
cd /tmp
echo "== shim, first call of a process WITHOUT the warm-up ==" > $OUT/r06_c_shim_incremental.txt
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py >> $OUT/r06_c_shim_incremental.txt 2>&1
echo "== shim, first call of a process AFTER sfmba_device_warmup(0, n_obs) ==" >> $OUT/r06_c_shim_incremental.txt
SFMBA_BUILD_TIMING=1 python $REPO/tools/time_shim_incremental.py --warmup >> $OUT/r06_c_shim_incremental.txt 2>&1
grep "warmup\|wall time\|path:" $OUT/r06_c_shim_incremental.txt
rm -rf $OUT/mk
SFMBA_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/mk -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --extra-workloads 0 > /dev/null 2> $OUT/mk.err
find $OUT/mk -name "*marker*" | head; f=$(find $OUT/mk -name "*marker_api_trace.csv" | head -1); [ -n "$f" ] && (head -3 "$f"; cut -d, -f3 "$f" | sort | uniq -c | sort -rn | head -8; cp "$f" $OUT/r06_c_cfg3_roctx_marker_trace.csv)
rm -rf $OUT/mk

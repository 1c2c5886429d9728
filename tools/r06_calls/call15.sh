#!/bin/bash
# round 6, GPU call 15: what k_sy_cg spends beyond k_sy_prod -- what-if builds (wrong results): no slice work, no partial-sum atomics, no atomics on w
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_15
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 3 --warmup 1 --opt max_iters=3"
for v in "" wi_noslice wi_nopart wi_nowat two; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err; fi
echo "== ${v:-default}"
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/stats_$v.txt "x" | grep "k_sy_cg\|k_sy_prod\|k_sy_vec" | cut -c1-60,110-170
rm -rf $OUT/st
done

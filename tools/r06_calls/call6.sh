#!/bin/bash
# round 6, GPU call 6: what bounds k_sy_coarse -- without its global atomics (wrong results), 256- and 64-row tiles
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_6
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 4 --warmup 1"
for v in "" co_noat co_r256 co_r64; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err; fi
echo "== ${v:-default}"
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/stats_$v.txt "x" | grep "k_sy_coarse\|k_sy_e\|coarse_invert" | cut -c1-60,110-170
rm -rf $OUT/st
done

#!/bin/bash
# round 6, GPU call 23: bisecting the 3 us between k_sy_prod (18.0) and the micro-benchmark kernel on the same matrix (15.0): no flag test / p through the scalar unit / no epilogue / all three
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_23
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0"
for v in wi_noflag wi_scalarp wi_noepi wi_all3; do
rm -rf $OUT/st
SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/st.err
echo "== $v"
python $REPO/tools/trace_seq.py $OUT/st 2>/dev/null | grep -A3 "k_sy_vec<true" | head -9
rm -rf $OUT/st
done

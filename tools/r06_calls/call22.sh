#!/bin/bash
# round 6, GPU call 22: what-if -- the micro-benchmark kernel (k_symv<8,true>) launched on the REAL matrix in front of k_sy_prod: 14.7 us there, what here?
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_22
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0"
rm -rf $OUT/st
SFMBA_LIB=$REPO/tools/ab/wi_copy/libsfmba_hip.so rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/st.err
python $REPO/tools/trace_seq.py $OUT/st 2>/dev/null | grep -A7 "k_sy_vec<true" | head -24
rm -rf $OUT/st

#!/bin/bash
# round 6, GPU call 21: what-if -- k_sy_prod launched three times back to back (results wrong): is the 17.9 us the kernel's, or what it inherits from the launch in front of it?
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_21
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0"
rm -rf $OUT/st
SFMBA_LIB=$REPO/tools/ab/wi_rep2/libsfmba_hip.so rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/st.err
python $REPO/tools/trace_seq.py $OUT/st 2>/dev/null | grep -A9 "k_sy_vec<true" | head -24
rm -rf $OUT/st

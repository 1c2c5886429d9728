#!/bin/bash
# round 6, GPU call 16: ordered kernel traces of one cfg-5 solve: one launch per CG iteration (k_sy_cg) against two (k_sy_vec + k_sy_prod)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_16
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 3 --warmup 2"
for v in "" two; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err; fi
echo "== ${v:-default}"
python $REPO/tools/trace_seq.py $OUT/st > $OUT/seq_${v:-one}.txt; head -42 $OUT/seq_${v:-one}.txt
rm -rf $OUT/st
done

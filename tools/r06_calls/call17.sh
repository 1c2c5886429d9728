#!/bin/bash
# round 6, GPU call 17: the symmetric-CG tests on the two-launch tree again; what k_sy_prod spends beyond the bare product (what-if builds, first launch of a solve in the ordered trace)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_17
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "symmetric or cfg5_real_unsharded or f32_matrix" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; grep -v "Ceres Solver Report" $OUT/tests.log | tail -4
cd /tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 2 --warmup 1"
for v in "" wi_noslice wi_nopart; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B > /dev/null 2> $OUT/st.err; fi
echo "== ${v:-default}"
python $REPO/tools/trace_seq.py $OUT/st | grep -A3 "k_sy_vec<true" | head -8
rm -rf $OUT/st
done

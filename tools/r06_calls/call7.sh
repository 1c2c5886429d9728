#!/bin/bash
# round 6, GPU call 7: k_sy_coarse specialised on the diagonal flag with the prefetch peeled; W~ of the lane's columns held narrow (variant: conversion kept opaque)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_7
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 4 --warmup 1"
for v in "" co_opaque; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > $OUT/b_$v.json 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/$v/libsfmba_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B > $OUT/b_$v.json 2> $OUT/st.err; fi
echo "== ${v:-default}"; python -c "
import json; d=json.loads([l for l in open('$OUT/b_$v.json') if l.startswith('{')][-1]); print(d['value'], d.get('parity_ok'))"
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/stats_$v.txt "x" | grep "k_sy_coarse\|k_sy_e\|coarse_invert\|k_sy_prod\|k_sy_vec<false" | cut -c1-60,110-170
rm -rf $OUT/st
done

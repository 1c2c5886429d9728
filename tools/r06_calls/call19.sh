#!/bin/bash
# round 6, GPU call 19: the symv micro-benchmark under rocprofv3 --kernel-trace --stats: real kernel durations of the tile forms (the event averages it prints subtract a memset)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_19
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/mb
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mb -- $REPO/tools/micro/symv_bench 6001 > $OUT/symv_print.txt 2> $OUT/mb.err
f=$(find $OUT/mb -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    print("%-110s calls %5s avg %9.2f us  min %9.2f  max %9.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
tail -5 $OUT/symv_print.txt
rm -rf $OUT/mb

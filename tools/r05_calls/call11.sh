#!/bin/bash
# ordered kernel traces of the sharded loops on one rank (which memsets / copies sit between the kernels of an LM iteration)
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py"
rm -rf $OUT/tr_row5 $OUT/tr_rep3
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_row5 -- $BENCH --mode sharded --workload cfg5 --row-sharded --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/tr_row5.err
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_rep3 -- $BENCH --mode sharded --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic > /dev/null 2> $OUT/tr_rep3.err
python - <<PY
import csv, glob
for tag in ("tr_row5", "tr_rep3"):
    f = glob.glob("$OUT/%s/**/*kernel_trace.csv" % tag, recursive=True)
    rows = []
    for p in f:
        rows += list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open("$OUT/%s_sequence.txt" % tag, "w") as o:
        prev_end = None
        for r in rows:
            name = r["Kernel_Name"].replace("sfmba::", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write("%-52s %9.2f us  gap %8.2f us\n" % (name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
            prev_end = e
    print(tag, len(rows))
PY
rm -rf $OUT/tr_row5 $OUT/tr_rep3
ls -la $OUT

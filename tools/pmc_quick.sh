#!/bin/bash
# one --pmc pass of the SQ issue counters over a short bench run (runs ON THE GPU BOX):  [ENV=..] tools/pmc_quick.sh [kernel substring] [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/pmc_quick; rm -rf $OUT; mkdir -p $OUT
K=${1:-k_schur_pairs}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic "$@" > /dev/null 2> $OUT/err.txt
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "$K" not in k: continue
        k=k.split("(")[0]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in acc:
    print(k)
    for c in sorted(acc[k]): print("    %-28s %14.0f"%(c, acc[k][c]/cnt[k][c]))
PY
rm -rf $OUT/p

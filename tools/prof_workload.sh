#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics + the two PMC traffic passes of ONE bench configuration -> $OUT/<tag>_<name>_{kernel_stats,pmc_traffic}.txt + bench line
#   tools/prof_workload.sh <outdir> <tag> <name> <bench args...>      e.g.  tools/prof_workload.sh gpurun_out/r06_1 r06_a cfg3_f64_auto --precision f64 --linear auto
set -u
OUT=$1; TAG=$2; NAME=$3; shift 3
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT
OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-live-traffic --extra-workloads 0 $*"
$BENCH --steps 10 --warmup 3 2> $OUT/${NAME}.err | grep '^{' > $OUT/${TAG}_${NAME}_bench.json
rm -rf $OUT/st_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$NAME -- $BENCH --steps 6 --warmup 2 > /dev/null 2>> $OUT/${NAME}.err
python $REPO/tools/rocprof_summary.py $OUT/st_$NAME $OUT/${TAG}_${NAME}_kernel_stats.txt "$TAG: bench.py $* --steps 6 --warmup 2 under rocprofv3 --kernel-trace --stats" > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${NAME}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${NAME}_$c -- $BENCH --steps 2 --warmup 1 > /dev/null 2>> $OUT/${NAME}.err
done
python $REPO/tools/pmc_summary.py $OUT/pmc_${NAME}_FETCH_SIZE $OUT/pmc_${NAME}_WRITE_SIZE $OUT/${TAG}_${NAME}_pmc_traffic.txt "python bench.py $* --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic" > /dev/null
rm -rf $OUT/st_$NAME $OUT/pmc_${NAME}_FETCH_SIZE $OUT/pmc_${NAME}_WRITE_SIZE
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_${NAME}_bench.json").read().strip().splitlines()[-1])
print("%-28s %9.1f LM it/s  %.4f ms/step  %s" % ("$NAME", d["value"], d["ms_per_step"], d.get("termination")))
PY

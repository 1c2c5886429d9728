#!/bin/bash
# kernel sequence (start, gap, duration) of the last solve of a short bench run:  tools/prof_seq.sh [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/seq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/st
rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic "$@" > $OUT/bench.json 2> $OUT/err.txt
python $REPO/tools/trace_seq.py $OUT/st > $OUT/seq.txt
rm -rf $OUT/st
cat $OUT/seq.txt

#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run, top kernels printed (runs ON THE GPU BOX): tools/quick_stats.sh [bench args]
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/quick_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic "$@" > $OUT/bench.json 2> $OUT/err.txt
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/stats.txt "quick" > /dev/null
rm -rf $OUT/st
head -24 $OUT/stats.txt

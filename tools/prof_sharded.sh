#!/bin/bash
# rocprofv3 kernel trace of ONE problem run through the sharded loop on one rank (RCCL 1-rank all-reduce): where the sharded mode's overhead goes
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/shprof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/st
rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- python $REPO/bench.py --mode sharded --steps 3 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/bench.json 2> $OUT/err.txt
python $REPO/tools/trace_seq.py $OUT/st > $OUT/seq.txt
rm -rf $OUT/st
head -150 $OUT/seq.txt

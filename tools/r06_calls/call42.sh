#!/bin/bash
# round 6, GPU call 42: the whole -m gpu suite on the final tree (padding rows, F32J radius after an invalid step, iteration limit 0), smoke, the default bench line
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_42
mkdir -p $OUT
cd $REPO; export TMPDIR=/tmp
for i in 1; do
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/gpu_suite_$i.log 2>&1
echo "suite $i rc=$?"; grep -v "Ceres Solver Report" $OUT/gpu_suite_$i.log | grep "passed\|failed\|FAILED\|Error" | tail -5
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "Ceres Solver Report" | tail -3
cd /tmp
python $REPO/bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_default.json
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read())
print('headline %.1f ms/step %.4f parity %s n_gpus %s' % (d['value'], d['ms_per_step'], d['parity_ok'], d['n_gpus']))"

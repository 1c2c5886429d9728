#!/bin/bash
# round 5, GPU call 7: the driver's commands as the driver runs them (smoke, default bench with its wall time, the sharded extras at one rank), then the whole GPU suite
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_7
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; cat $OUT/bench_default.time
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print('headline', round(d['value'],1), d['ms_per_step'], d['dtype'][:60]); print('roofline', {k:d['roofline'][k] for k in ('kernel','frac','traffic','traffic_is_static')}); print('cpu', d['cpu_baseline']['value'])
for k,v in d.get('extra_workloads',{}).items(): print(k, {a:v.get(a) for a in ('value','parity_ok','error')})"
( time python bench.py --sharded-extras 1 --steps 5 --no-cpu-baseline --no-live-traffic > $OUT/bench_extras.json 2> $OUT/bench_extras.err ) 2> $OUT/bench_extras.time; cat $OUT/bench_extras.time
python -c "
import json; d=json.load(open('$OUT/bench_extras.json'))
for k,v in d.get('sharded',{}).items(): print(k, {a:v.get(a) for a in ('value','ms_per_step','final_cost','error')} if isinstance(v,dict) else v)"
timeout -k 5 1200 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | tail -8

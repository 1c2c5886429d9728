#!/bin/bash
# round 5, GPU call 4: the 4-rank row_wide case with repeats under a watchdog, the whole GPU suite, quick benches, the shim
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_4
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
DBG_REPEATS=3 timeout -k 5 200 python tools/debug_row_hang.py 4 row_wide 90 > $OUT/debug_row_hang.log 2>&1
grep -v "amdgpu.ids\|hostname of the client" $OUT/debug_row_hang.log | tail -70
for r in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('cfg3', round(d['value'],1), 'it/s', d['ms_per_step'], {n: k[n] for n in k}, d['default_solver_auto']['value'])" >> $OUT/bench_quick.txt 2>&1
done
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('cfg5', round(d['value'],1), 'it/s', d['ms_per_step'], {n: k[n] for n in k})" >> $OUT/bench_quick.txt 2>&1
cat $OUT/bench_quick.txt
timeout -k 5 900 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; tail -25 $OUT/gpu_suite.log
( echo "== SFMBA_LINEAR=pcg =="; SFMBA_LINEAR=pcg timeout 200 python tools/time_shim_incremental.py; echo "== default (auto) =="; timeout 200 python tools/time_shim_incremental.py ) > $OUT/shim_incremental.txt 2>&1
grep "wall time\|path:\|marshal split" $OUT/shim_incremental.txt

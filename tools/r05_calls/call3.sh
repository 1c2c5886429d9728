#!/bin/bash
# round 5, GPU call 3: the 4-rank row_wide hang under a watchdog, the tests of what changed since call 2, quick benches, the AUTO float sweep, the shim
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_3
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 150 python tools/debug_row_hang.py 4 row_wide 50 > $OUT/debug_row_hang.log 2>&1
grep -v "amdgpu.ids\|hostname of the client" $OUT/debug_row_hang.log | tail -60
timeout -k 5 420 python -m pytest tests/test_gpu_options_v4.py tests/test_gpu_parity.py tests/test_gpu_matrix_free.py tests/test_gpu_fullsize.py tests/test_gpu_shim.py tests/test_gpu_append.py -x -q --timeout 300 > $OUT/tests_a.log 2>&1
echo "tests_a rc=$?" >> $OUT/tests_a.log; tail -15 $OUT/tests_a.log
for r in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('cfg3', round(d['value'],1), 'it/s', d['ms_per_step'], {n: k[n] for n in k})" >> $OUT/bench_quick.txt 2>&1
done
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('cfg5', round(d['value'],1), 'it/s', d['ms_per_step'], {n: k[n] for n in k})" >> $OUT/bench_quick.txt 2>&1
cat $OUT/bench_quick.txt
timeout 300 python tools/auto_float_sweep.py > $OUT/auto_float_sweep.txt 2>&1; cat $OUT/auto_float_sweep.txt | grep -v amdgpu
( echo "== SFMBA_LINEAR=pcg =="; SFMBA_LINEAR=pcg timeout 200 python tools/time_shim_incremental.py; echo "== default (auto) =="; timeout 200 python tools/time_shim_incremental.py ) > $OUT/shim_incremental.txt 2>&1
grep -v amdgpu $OUT/shim_incremental.txt | tail -40

#!/bin/bash
# round 5, GPU call 1: the new paths first (row-sharded, matrix-free, RCCL bindings), then the A/B of the pair-pass accumulation, then the bench line with its extras
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r05_1
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_matrix_free.py tests/test_gpu_sharded.py -x -q --timeout 600 -k "matrix_free or pair_limit or no_pair or two_to_the or row_ or rccl_bindings or dist_cfg2 or dist_wide or dist_imp_cfg2 or duplicate" > $OUT/new_tests.log 2>&1
echo "new tests rc=$?" >> $OUT/new_tests.log
tail -30 $OUT/new_tests.log
timeout 600 python -m pytest tests/test_gpu_options_v4.py tests/test_gpu_pair_forms.py tests/test_gpu_baseline_parity.py -x -q --timeout 600 > $OUT/affected_tests.log 2>&1
echo "affected rc=$?" >> $OUT/affected_tests.log
tail -5 $OUT/affected_tests.log
for v in cur acc0 acc2; do
  LIB=$REPO/sfm-toy-library_amd/csrc/libsfmba_hip.so
  [ $v != cur ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  for r in 1 2; do
    SFMBA_LIB=$LIB timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('$v cfg3', round(d['value'],1), 'it/s', 'pairs %.1f' % k.get('schur_pairs', 0), 'rms %.9f' % d['final_rms_px'], 'cost %.6f' % d['final_cost'])" >> $OUT/ab_pair_acc.txt 2>&1
  done
  SFMBA_LIB=$LIB timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --extra-workloads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']; print('$v cfg5', round(d['value'],1), 'it/s', 'pairs %.1f' % k.get('schur_pairs', 0), 'rms %.9f' % d['final_rms_px'], 'cost %.6f' % d['final_cost'])" >> $OUT/ab_pair_acc.txt 2>&1
done
cat $OUT/ab_pair_acc.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print('headline', d['value'], d['ms_per_step']); 
for k,v in d.get('extra_workloads',{}).items(): print(k, {a:v.get(a) for a in ('value','ms_per_step','parity_ok','rel_cost_diff_vs_oracle','error')})"
timeout 600 python bench.py --mode sharded --workload cfg5 --row-sharded --steps 5 --no-cpu-baseline --no-live-traffic 2> $OUT/sh_row5.err | grep '^{' > $OUT/cfg5_row_sharded_1rank.json
timeout 600 python bench.py --mode sharded --workload cfg5 --distributed-cg --steps 5 --no-cpu-baseline --no-live-traffic 2> $OUT/sh_d5.err | grep '^{' > $OUT/cfg5_dist_1rank.json
timeout 300 python bench.py --mode sharded --workload cfg3 --row-sharded --steps 10 --no-cpu-baseline --no-live-traffic 2> $OUT/sh_row3.err | grep '^{' > $OUT/cfg3_row_sharded_1rank.json
python -c "
import json
for f in ('cfg5_row_sharded_1rank','cfg5_dist_1rank','cfg3_row_sharded_1rank'):
    try:
        d=json.load(open('$OUT/'+f+'.json')); print(f, round(d['value'],1), d['ms_per_step'], d['sharded'].get('one_rank_without_collective',{}).get('value'), d['final_cost'])
    except Exception as e: print(f, 'failed', e)"

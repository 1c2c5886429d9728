cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_options_v4.py -q -x -k "imp or switches or dist" 2>&1 | grep -v "Ceres Solver Report" | tail -15
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py -q -x -k "cfg5_real" 2>&1 | grep -v "Ceres Solver Report" | tail -8
cd /tmp && export TMPDIR=/tmp
for v in "" "--distributed-cg" "--implicit-cg"; do
  for wl in cfg3 cfg5; do
  echo "== sharded $wl $v"; python $GRAFT_REPO_ROOT/bench.py --mode sharded --workload $wl --steps 5 --no-cpu-baseline --no-live-traffic $v 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['sharded']['linear_iters_per_step'], d['final_cost'])"
  done
done
rm -rf /tmp/imp_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/imp_stats -- python $GRAFT_REPO_ROOT/bench.py --mode sharded --workload cfg5 --steps 3 --no-cpu-baseline --no-live-traffic --implicit-cg > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/imp_stats $GRAFT_REPO_ROOT/gpurun_out/r04_b_cfg5_sharded_1rank_implicit_cg_kernel_stats.txt "r04_b: bench.py --mode sharded --workload cfg5 --steps 3 --implicit-cg (one rank, f32j) under rocprofv3 --kernel-trace --stats" | cut -c1-150 | head -20

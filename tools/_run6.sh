cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print(round(d["value"],1), round(d["ms_per_step"],4), d["final_cost"], d["kernel_profile_us"])'
for wl in cfg3 cfg5 cfg3_banded; do
  st=20; [ $wl = cfg5 ] && st=5
  for g in 0 1; do
    echo "== $wl SFMBA_PB_GROUP=$g"; SFMBA_PB_GROUP=$g python $R/bench.py --workload $wl --no-cpu-baseline --no-live-traffic --steps $st --warmup 3 2>/dev/null | python -c "$pick"
  done
done
cd $R && SFMBA_PB_GROUP=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pair_forms.py tests/test_gpu_random.py tests/test_gpu_rejections.py -q -x 2>&1 | grep -v "Ceres Solver Report" | tail -15

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print(round(d["value"],1), round(d["ms_per_step"],4), d["final_cost"], d["kernel_profile_us"]["schur_pairs"])'
for wl in cfg5; do
  for lib in "" tools/ab/libsfmba_subf4.so; do
  echo "== $wl $lib"; SFMBA_LIB=${lib:+$R/$lib} python $R/bench.py --workload $wl --no-cpu-baseline --no-live-traffic --steps 5 --warmup 3 2>/dev/null | python -c "$pick"
  done
done
cd $R && timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Ceres Solver" | tail -5

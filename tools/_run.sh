cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print(round(d["value"],1), round(d["ms_per_step"],4), d["final_cost"], d["kernel_profile_us"]["cam_diag"])'
for wl in cfg3 cfg5; do
  st=20; [ $wl = cfg5 ] && st=5
  for lib in "" tools/ab/libsfmba_chunk512.so tools/ab/libsfmba_chunk1024.so; do
  echo "== $wl $lib"; SFMBA_LIB=${lib:+$R/$lib} python $R/bench.py --workload $wl --no-cpu-baseline --no-live-traffic --steps $st --warmup 3 2>/dev/null | python -c "$pick"
  done
done

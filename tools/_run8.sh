cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pick='import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1]); print(round(d["value"],1), round(d["ms_per_step"],4), d["final_cost"], d["kernel_profile_us"])'
export SFMBA_PB_GROUP=1 SFMBA_PU_GROUP=1
for wl in cfg3_banded; do
    echo "== $wl grouped + sorted points"; python $R/bench.py --workload $wl --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 2>/dev/null | python -c "$pick"
done
cd $R && timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Ceres Solver Report" | tail -15

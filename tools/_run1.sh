cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --steps 5 --warmup 2"
for v in "SFMBA_PAIR_FORM=0" "SFMBA_SUBF_WAVES=3" "SFMBA_SUBF_WAVES=4"; do
  echo "== $v"; env $v $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_cost'], d['kernel_profile_us'])"
done
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15

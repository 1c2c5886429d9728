cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sh_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh_stats -- python $R/bench.py --mode sharded --workload cfg3 --steps 20 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
python $R/tools/rocprof_summary.py /tmp/sh_stats /tmp/sh.txt "sharded cfg3 1 rank" | cut -c1-60,108-160 | head -45
python $R/bench.py --workload cfg3 --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('unsharded', round(d['value'],1), round(d['ms_per_step'],4))"

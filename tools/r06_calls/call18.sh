#!/bin/bash
# round 6, GPU call 18: the whole -m gpu suite on the two-launch tree (k_sy_cg removed), bench watchdog moved in front of the RCCL count; default bench line + a sharded-extras run on one rank
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_18
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 2700 python -m pytest tests -m gpu -q --timeout 900 > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?" >> $OUT/gpu_suite.log; grep -v "Ceres Solver Report" $OUT/gpu_suite.log | grep "passed\|failed\|FAILED\|Error\|rc=" | tail -12
cd /tmp
python $REPO/bench.py 2> $OUT/bench.err | tail -1 > $OUT/r06_d_cfg3_pcg_bench.json
python - <<PY
import json
d=json.loads(open("$OUT/r06_d_cfg3_pcg_bench.json").read())
print("headline %.1f  ms/step %.4f parity %s" % (d["value"], d["ms_per_step"], d.get("parity_ok")))
for k,v in d.get("extra_workloads",{}).items(): print("  %-24s %9.1f %s %s" % (k, v.get("value",0), v.get("parity_ok"), v.get("error","")[:100]))
print(d["kernel_profile_us"])
PY
python $REPO/bench.py --sharded-extras 1 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 5 --warmup 2 2> $OUT/bench_sh.err | tail -1 > $OUT/bench_sharded_one_rank.json
python - <<PY
import json
d=json.loads(open("$OUT/bench_sharded_one_rank.json").read())
print("sharded-extras line: headline %.1f n_gpus %s abandoned %s" % (d["value"], d["n_gpus"], d.get("abandoned")))
for k,v in d.get("sharded",{}).items(): print("  %-28s %9.1f %s %s" % (k, v.get("value",0), v.get("parity_ok"), str(v.get("error",""))[:100]))
PY

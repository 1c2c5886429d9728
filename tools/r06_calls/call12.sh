#!/bin/bash
# round 6, GPU call 12: the whole -m gpu suite on the tree with the ADVICE r5 changes, ||x|| folded into the column-norm pass, the large-camera tests; default bench line
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_12
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

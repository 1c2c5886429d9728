#!/bin/bash
# round 6, GPU call 1: this round's baseline on a fresh box (default bench line) + the evidence VERDICT r5 item 5 asks for:
# kernel statistics and counter traffic of what a drop-in caller runs (fp64 + AUTO) at cfg 3, cfg 2 with the reference's solver, cfg3_banded default
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py 2> $OUT/bench.err | tail -1 > $OUT/r06_a_cfg3_pcg_bench.json
python - <<PY
import json
d=json.loads(open("$OUT/r06_a_cfg3_pcg_bench.json").read())
print("headline %.1f  ms/step %.4f" % (d["value"], d["ms_per_step"]))
for k,v in d.get("extra_workloads",{}).items(): print("  %-24s %9.1f %s" % (k, v.get("value",0), v.get("parity_ok")))
PY
bash $REPO/tools/prof_workload.sh $OUT r06_a cfg3_f64_auto --precision f64 --linear auto
bash $REPO/tools/prof_workload.sh $OUT r06_a cfg3_f64_pcg --precision f64 --linear pcg
bash $REPO/tools/prof_workload.sh $OUT r06_a cfg2_f64_dense_schur --workload cfg2 --precision f64 --linear cholesky
bash $REPO/tools/prof_workload.sh $OUT r06_a cfg3_banded_auto --workload cfg3_banded --linear auto
bash $REPO/tools/prof_workload.sh $OUT r06_a cfg3_banded_f64_auto --workload cfg3_banded --precision f64 --linear auto
ls $OUT

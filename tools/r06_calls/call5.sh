#!/bin/bash
# round 6, GPU call 5: the coarse set-up on the one triangle (k_sy_coarse + k_sy_e) and the pair pass writing the upper triangle only: streaming-path tests,
# cfg 5 A/B, kernel statistics + counter traffic of cfg 5
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_5
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "cfg5 or f32_matrix or fullsize or segments or matrix_free or options_v4 or sharded" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; grep -v "Ceres Solver Report" $OUT/tests.log | tail -15
cd /tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 6 --warmup 2"
run() { name=$1; shift; "$@" 2> $OUT/$name.err | grep '^{' > $OUT/$name.json; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read()); k=d["kernel_profile_us"]
print("%-22s %8.1f it/s  %.3f ms/step  pcg_iter %.2f us  setup %.2f  pairs %.1f  parity %s" % ("$name", d["value"], d["ms_per_step"], k.get("pcg_iter",0), k.get("pcg_setup",0), k.get("schur_pairs",0), d.get("parity_ok")))
PY
}
run cfg5_sym $B
run cfg5_full $B --opt pcg_symmetric=-1
run cfg5_sym_again $B
run cfg5_sym_auto $B --linear auto
bash $REPO/tools/prof_workload.sh $OUT r06_b cfg5_pcg --workload cfg5
head -22 $OUT/r06_b_cfg5_pcg_kernel_stats.txt | cut -c1-150
head -12 $OUT/r06_b_cfg5_pcg_pmc_traffic.txt

#!/bin/bash
# round 6, GPU call 2: the symmetric streaming CG (k_pcg_iter_sym), first run: the tests that reach the streaming path, cfg 5 A/B against the round-5 kernel
# (--opt pcg_symmetric=-1) and the compile-time variants (waves per SIMD, workgroups), kernel statistics of cfg 5
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_2
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "cfg5 or f32_matrix or fullsize or segments or matrix_free or options_v4" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; grep -v "Ceres Solver Report" $OUT/tests.log | tail -15
cd /tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 6 --warmup 2"
run() { name=$1; shift; "$@" 2> $OUT/$name.err | grep '^{' > $OUT/$name.json; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read()); k=d["kernel_profile_us"]
print("%-22s %8.1f it/s  %.3f ms/step  pcg_iter %.2f us  setup %.2f  pairs %.1f  parity %s  cg/step %s" % ("$name", d["value"], d["ms_per_step"], k.get("pcg_iter",0), k.get("pcg_setup",0), k.get("schur_pairs",0), d.get("parity_ok"), d["config"].get("lm_iterations_per_step")))
PY
}
run cfg5_sym $B
run cfg5_full $B --opt pcg_symmetric=-1
SFMBA_LIB=$REPO/tools/ab/sy_w3/libsfmba_hip.so run cfg5_sym_w3 $B
SFMBA_LIB=$REPO/tools/ab/sy_wg512/libsfmba_hip.so run cfg5_sym_wg512 $B
SFMBA_LIB=$REPO/tools/ab/sy_wg768_w3/libsfmba_hip.so run cfg5_sym_wg768_w3 $B
run cfg5_sym_again $B
rm -rf $OUT/st
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/st.err
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/r06_b_cfg5_pcg_kernel_stats.txt "r06_b: bench.py --workload cfg5 --steps 4 --warmup 1 (f32j, PCG, symmetric streaming CG) under rocprofv3 --kernel-trace --stats" | head -16
rm -rf $OUT/st

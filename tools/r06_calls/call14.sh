#!/bin/bash
# round 6, GPU call 14: ONE launch per CG iteration on the triangle (k_sy_cg, Chronopoulos - Gear recurrences): the streaming-path tests, cfg 5 A/B against the
# two-launch form (-DSFMBA_SY_TWO_LAUNCH) and the round-5 kernels, kernel statistics
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_14
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests -m gpu -q --timeout 600 -x -k "symmetric or cfg5 or f32_matrix or fullsize or matrix_free or options_v4 or large_cameras" > $OUT/tests.log 2>&1
echo "tests rc=$?" >> $OUT/tests.log; grep -v "Ceres Solver Report" $OUT/tests.log | tail -25
cd /tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 6 --warmup 2"
run() { name=$1; shift; "$@" 2> $OUT/$name.err | grep '^{' > $OUT/$name.json; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read()); k=d["kernel_profile_us"]
print("%-22s %8.1f it/s  %.3f ms/step  pcg_iter %.2f us  setup %.2f  pairs %.1f  parity %s  cg its/step %.1f" % ("$name", d["value"], d["ms_per_step"], k.get("pcg_iter",0), k.get("pcg_setup",0), k.get("schur_pairs",0), d.get("parity_ok"), d.get("whole_iteration_hbm",{}).get("algorithmic_bytes_per_iteration",0) and 0))
PY
}
run cfg5_one_launch $B
SFMBA_LIB=$REPO/tools/ab/sy_two_launch/libsfmba_hip.so run cfg5_two_launch $B
run cfg5_full $B --opt pcg_symmetric=-1
run cfg5_one_launch_auto $B --linear auto
SFMBA_LIB=$REPO/tools/ab/sy_two_launch/libsfmba_hip.so run cfg5_two_launch_auto $B --linear auto
run cfg5_one_launch_f64 $B --precision f64
rm -rf $OUT/st
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/st.err
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/r06_f_cfg5_pcg_kernel_stats.txt "r06_f: bench.py --workload cfg5 --steps 4 --warmup 1 (f32j, PCG, k_sy_cg) under rocprofv3 --kernel-trace --stats" | head -14 | cut -c1-72,112-160
rm -rf $OUT/st

#!/bin/bash
# round 6, GPU call 4: k_sy_vec with every load of the launch in flight before its first reduction; what-if: the pair pass writing the upper triangle only
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_4
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0 --steps 6 --warmup 2"
run() { name=$1; shift; "$@" 2> $OUT/$name.err | grep '^{' > $OUT/$name.json; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read()); k=d["kernel_profile_us"]
print("%-22s %8.1f it/s  %.3f ms/step  pcg_iter %.2f us  setup %.2f  pairs %.1f  parity %s" % ("$name", d["value"], d["ms_per_step"], k.get("pcg_iter",0), k.get("pcg_setup",0), k.get("schur_pairs",0), d.get("parity_ok")))
PY
}
run cfg5_sym $B
SFMBA_LIB=$REPO/tools/ab/whatif_upper/libsfmba_hip.so run cfg5_whatif_upper $B
run cfg5_sym_again $B
for v in "" whatif; do
rm -rf $OUT/st
if [ -z "$v" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/st.err
else SFMBA_LIB=$REPO/tools/ab/whatif_upper/libsfmba_hip.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- $B --steps 4 --warmup 1 > /dev/null 2> $OUT/st.err; fi
python $REPO/tools/rocprof_summary.py $OUT/st $OUT/stats_$v.txt "x" | grep "k_schur_pairs_sub\|k_sy_\|k_pcg_coarse" 
rm -rf $OUT/st
done

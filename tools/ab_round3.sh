#!/bin/bash
# Runs ON THE GPU BOX: A/B of the round-3 re-evaluating reduced-system passes against the record-gathering ones (cfg 3, bench mode).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3"
pick='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], "it/s %.1f" % d["value"], "ms/step %.4f" % d["ms_per_step"], {k: v for k, v in d["kernel_profile_us"].items()}, "auto:", d.get("default_solver_auto", {}).get("value"), d.get("default_solver_auto", {}).get("ms_per_step"), d.get("default_solver_auto", {}).get("cg_iterations_per_step"))'
SFMBA_SCHUR_RECORDS=1 $B 2>$OUT/rec.err | python -c "$pick" records | tee $OUT/ab.txt
SFMBA_PAIR_FORM=3 $B 2>$OUT/rc3.err | python -c "$pick" recompute_unfactored | tee -a $OUT/ab.txt
$B 2>$OUT/rc4.err | python -c "$pick" recompute_factored | tee -a $OUT/ab.txt
$B --linear auto 2>$OUT/auto.err | python -c "$pick" auto | tee -a $OUT/ab.txt
$B --linear cholesky 2>$OUT/chol.err | python -c "$pick" cholesky | tee -a $OUT/ab.txt
$B --workload cfg3_banded 2>$OUT/banded.err | python -c "$pick" banded_recompute | tee -a $OUT/ab.txt
SFMBA_SCHUR_RECORDS=1 $B --workload cfg3_banded 2>$OUT/banded_rec.err | python -c "$pick" banded_records | tee -a $OUT/ab.txt
tail -3 $OUT/*.err

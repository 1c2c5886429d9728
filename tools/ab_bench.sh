#!/bin/bash
# A/B of an environment switch on the headline bench: tools/ab_bench.sh VAR "0 1" [reps]
VAR=$1; VALS=$2; REPS=${3:-3}
for r in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms/solve', d['kernel_profile_us'].get('pcg_iter'), d['kernel_profile_us'].get('schur_pairs'), d['kernel_profile_us'].get('finalize'))"
done; done

#!/bin/bash
# A/B of an environment switch on the headline bench: tools/ab_bench.sh VAR "0 1" [reps] [kernel names of kernel_profile_us to print]
VAR=$1; VALS=$2; REPS=${3:-3}; KERN=${4:-"pcg_iter schur_pairs finalize point_build point_update cam_diag"}
for r in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic | KERN="$KERN" python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); k=d['kernel_profile_us']
print('$VAR=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms/solve', ' '.join('%s %.1f' % (n, k.get(n, float('nan'))) for n in os.environ['KERN'].split()))"
done; done

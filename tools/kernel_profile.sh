#!/bin/bash
# Per-kernel HIP-event averages of the headline bench under an environment switch: tools/kernel_profile.sh VAR "0 1"
VAR=${1:-SFMBA_NONE}; VALS=${2:-0}
for v in $VALS; do
  env $VAR=$v python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms/solve')
print('   ', {k:round(v,2) for k,v in d['kernel_profile_us'].items()})
print('   ', {k:round(v,3) for k,v in d['kernel_profile_share'].items()})"
done

#!/bin/bash
# round 6, GPU call 20: k_sy_prod -- tile descriptor fetched with the flag (two scalar round trips less), an unmasked path for interior tiles, occupancy variants: ordered trace + cfg 5 bench each
set -u
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r06_20
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py --workload cfg5 --no-cpu-baseline --no-live-traffic --extra-workloads 0"
for v in "" nointerior lb5 lb4 nointerior_lb5; do
  L=""; [ -n "$v" ] && L=$REPO/tools/ab/$v/libsfmba_hip.so
  rm -rf $OUT/st
  SFMBA_LIB=$L rocprofv3 --kernel-trace --output-format csv -d $OUT/st -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/st.err
  echo "== ${v:-default}"
  python $REPO/tools/trace_seq.py $OUT/st 2>/dev/null | grep -A5 "k_sy_vec<true" | head -6
  rm -rf $OUT/st
  for i in 1 2; do SFMBA_LIB=$L $B --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench %.1f it/s parity %s' % (d['value'], d['parity_ok']))"; done
done

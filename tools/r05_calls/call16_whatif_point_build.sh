cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r05_16; mkdir -p $OUT
for wl in cfg5 cfg3; do for v in default pb4; do
  LIB=""; [ $v != default ] && LIB=$REPO/tools/ab/$v/libsfmba_hip.so
  rm -rf $OUT/st
  SFMBA_LIB=$LIB timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python $REPO/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --extra-workloads 0 > /dev/null 2> $OUT/st.err
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_point_update" in r["Name"] or "k_point_build" in r["Name"]:
            print("%-5s %-8s %-30s calls %5s avg %9.2f us" % ("$wl", "$v", r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
rm -rf $OUT/st

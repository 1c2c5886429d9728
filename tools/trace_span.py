"""Per-solve timeline statistics from a rocprofv3 --kernel-trace CSV: span, busy time, gap histogram."""
import csv, sys, glob, os
path = sys.argv[1]
f = glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
names = [r["Kernel_Name"] for r in rows]
# solves are delimited by k_cam_setup (first kernel of the per-solve setup)
starts = [i for i, n in enumerate(names) if "k_cam_setup" in n]
starts.append(len(rows))
for a, b in list(zip(starts[:-1], starts[1:]))[-3:]:
    span = (en[b - 1] - st[a]) / 1e3
    busy = sum(en[i] - st[i] for i in range(a, b)) / 1e3
    gaps = [(st[i] - en[i - 1]) / 1e3 for i in range(a + 1, b)]
    big = sorted(gaps)[-6:]
    print("kernels %3d  span %8.1f us  busy %8.1f us  gaps total %7.1f  median %.2f  largest %s" % (b - a, span, busy, sum(gaps), sorted(gaps)[len(gaps) // 2], [round(g, 1) for g in big]))

"""Largest inter-kernel gaps of one LM solve from a rocprofv3 kernel trace, with the kernels on either side."""
import csv, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfmba::", "")[:26]
# last solve: from the last k_xnorm to the end
idx = [i for i, r in enumerate(rows) if "k_xnorm" in r["Kernel_Name"]]
a = idx[-2] - 2; b = idx[-1] - 2
gaps = []
for i in range(a + 1, b):
    g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
    gaps.append((g, nm(rows[i - 1]), nm(rows[i])))
print("kernels %d, span %.1f us, busy %.1f us, gaps %.1f us" % (b - a, (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3,
      sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3, sum(g for g, _, _ in gaps)))
import collections
by = collections.defaultdict(list)
for g, p, n in gaps: by[(p, n)].append(g)
for (p, n), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-28s -> %-28s n=%3d  mean %.2f us  total %.1f us" % (p, n, len(v), sum(v) / len(v), sum(v)))

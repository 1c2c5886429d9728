"""Kernel sequence (name, start offset, duration, gap before) of the LAST solve in a rocprofv3 kernel trace."""
import csv, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sfmba::", "")[:40]
idx = [i for i, r in enumerate(rows) if "k_begin" in r["Kernel_Name"]]
a = idx[-1]; b = len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, nm(r)))
    prev = e

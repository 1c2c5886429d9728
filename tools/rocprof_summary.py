"""Turns a rocprofv3 result (sqlite .db of --kernel-trace --stats, or *_kernel_stats.csv) into the
small text summary that is committed under profiles/."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    # durations are in ns in rocpd views when > 1e6 ... detect unit by magnitude of the smallest average
    return [dict(name=r[0], calls=int(r[1]), total=float(r[2]), avg=float(r[3]), pct=float(r[4])) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append(dict(name=r["Name"], calls=int(r["Calls"]), total=float(r["TotalDurationNs"]) / 1e3,
                            avg=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"])))
    return out


def main():
    src, dst, title = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    if os.path.isdir(src):
        dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
        csvs = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
        rows = from_csv(csvs[0]) if csvs else from_db(dbs[0])
    else:
        rows = from_csv(src) if src.endswith(".csv") else from_db(src)
    rows.sort(key=lambda r: -r["total"])
    with open(dst, "w") as f:
        f.write("# %s\n# source: rocprofv3 --kernel-trace --stats ; durations in microseconds\n" % title)
        f.write("%-110s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in rows:
            f.write("%-110s %8d %14.1f %12.3f %8.2f\n" % (r["name"][:110], r["calls"], r["total"], r["avg"], r["pct"]))
    print(open(dst).read())


if __name__ == "__main__":
    main()

"""Per-kernel average FETCH_SIZE / WRITE_SIZE (KB per launch) from two rocprofv3 --pmc passes (CSV output).
usage: pmc_summary.py <fetch_dir> <write_dir> <out.txt> "<command line that was profiled>" """
import csv, glob, os, sys, collections


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for path in f:
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[name] += float(r["Counter_Value"]); cnt[name] += 1
    return {k: acc[k] / cnt[k] for k in acc}


def main():
    fd, wd, out, cmd = sys.argv[1:5]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    names = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, 0) + wr.get(k, 0)))
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only)\n# command: %s\n" % cmd)
        f.write("# per-launch averages in KB as rocprofv3 reports them. gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests as 64 B\n"
                "# for wide coalesced streams (x2 correction applied by bench.py for the streaming kernels) and includes Infinity-Cache hits; WRITE_SIZE uncalibrated.\n")
        f.write("%-60s %16s %16s\n" % ("kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB"))
        for k in names:
            f.write("%-60s %16.1f %16.1f\n" % (k[:60], fe.get(k, 0.0), wr.get(k, 0.0)))
    print(open(out).read())


if __name__ == "__main__":
    main()

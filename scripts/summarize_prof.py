"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text table."""
import csv, glob, os, sys, collections
out = sys.argv[1]

def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None

def short(n):
    n = n.split("(")[0]
    for a, b in (("void ", ""), ("HIP_vector_type<float, 2u>", "cplx")):
        n = n.replace(a, b)
    return n[:60]

f = find("stats", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(f)))
    print("%-62s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:14]:
        print("%-62s %8s %12.3f %12.2f %7s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub in ("fetch", "write", "sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    print("== PMC pass '%s': per-dispatch averages ==" % sub)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:8]:
        n = max(1, len(cnt[k]))
        print("%-62s n=%d  " % (k, n) + "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))

"""per-kernel means of every counter in a rocprofv3 --pmc output directory (counter_collection.csv + kernel_trace.csv)"""
import csv, glob, os, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); us = collections.defaultdict(float)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"]); us[k] += dur.get(r["Dispatch_Id"], (k, 0))[1]
names = sorted({c for k in acc for c in acc[k]})
print("%-60s %6s %10s " % ("kernel", "calls", "avg_us") + " ".join("%22s" % n[:22] for n in names))
for k in sorted(acc, key=lambda k: -us[k])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = len(cnt[k])
    print("%-60s %6d %10.1f " % (k[:60], n, us[k] / n) + " ".join("%22.4g" % (acc[k].get(c, 0) / n) for c in names))

#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per launch)."""
import csv, glob, collections, sys
d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "nsr::" not in k: continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
lines = []
for k, v in sorted(agg.items()):
    lines.append(k)
    for c, x in sorted(v.items()):
        lines.append("    %-34s %14.4g   (mean of %d launches)" % (c, x / cnt[k][c], cnt[k][c]))
txt = "\n".join(lines)
print(txt)
if out: open(out, "a").write(txt + "\n")

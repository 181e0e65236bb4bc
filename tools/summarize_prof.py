#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel-trace stats + pmc passes) per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = n.replace("s2sr::", "").replace("void ", "")
    return n[:70]


for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"  {short(r['Name']):70s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {r['AverageNs']:>12s} pct {r['Percentage']}")

for p in sorted(glob.glob(f"{out}/pmc_*/")):
    files = glob.glob(f"{p}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    print("== pmc", p)
    for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
        print(" ", k)
        for c, v in agg[k].items():
            print(f"      {c:28s} sum {v:16.0f}  per-dispatch {v / max(cnt[k][c], 1):14.0f}  (n={cnt[k][c]})")

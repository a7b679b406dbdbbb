#!/usr/bin/env python
"""Condense rocprofv3 CSVs (kernel stats + PMC passes) into the few lines that get
committed under profiles/.  Usage: summarize_profile.py <dir with kt/ pmc_fetch/ pmc_write/>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("kt", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(f"{'calls':>8} {'avg_us':>12} {'total_ms':>12} {'pct':>7}  name")
    for r in rows[:18]:
        print(f"{r['Calls']:>8} {float(r['AverageNs']) / 1e3:12.3f} {float(r['TotalDurationNs']) / 1e6:12.3f} "
              f"{float(r['Percentage']):7.2f}  {r['Name'][:110]}")

for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = find(sub, "*counter_collection.csv")
    if not files:
        print(f"== {counter}: no counter_collection.csv found ==")
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    print(f"== {counter} per dispatch (raw counter units: KiB per rocprofv3; gfx950 FETCH_SIZE reads 1/2 of a wide "
          "coalesced stream -- MI355X_MICROARCH.md HBM section) ==")
    for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"{n:>8} dispatches  avg {tot / n:14.1f} KiB/dispatch  {k[:100]}")

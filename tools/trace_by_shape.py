#!/usr/bin/env python
"""Per-kernel average durations of a rocprofv3 kernel trace, split by launch SHAPE (grid size) and by duration class,
for a run in which one kernel template is launched at several sizes (the driver's bench command launches the CartPole
rollout at 65 536 x 250, 65 536 x 1 000 and 8 192 x 250).  Also the gap between consecutive dispatches of the headline
shape (launch-to-launch period - kernel duration).   Usage: trace_by_shape.py <dir with kt/>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rollout_staged" in n or "brax_kernel<1" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
rows.sort()
groups = defaultdict(list)
for k, (s, e, n, g, w) in enumerate(rows):
    short = n.split("(")[0].replace("void ", "").replace("carl::", "")[:70]
    groups[(short, g // w)].append((s, e, k))
print(f"{'kernel':72s} {'workgroups':>10s} {'class':>8s} {'calls':>6s} {'avg us':>9s} {'min':>8s} {'max':>8s} {'median gap to next same-shape dispatch us':>12s}")
for (short, wgs), lst in sorted(groups.items()):
    durs = sorted((e - s) / 1e3 for s, e, _ in lst)
    med = durs[len(durs) // 2]
    dur = lambda x: (x[1] - x[0]) / 1e3  # noqa: E731
    for label, sel in (("<med/2", [x for x in lst if dur(x) < 0.5 * med]), ("~median", [x for x in lst if 0.5 * med <= dur(x) <= 2 * med]),
                       (">2xmed", [x for x in lst if dur(x) > 2 * med])):
        if not sel:
            continue
        d = [(e - s) / 1e3 for s, e, _ in sel]
        gaps = sorted((sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 1) if sel[i + 1][2] == sel[i][2] + 1)
        g = gaps[len(gaps) // 2] if gaps else float("nan")
        print(f"{short:72s} {wgs:10d} {label:>8s} {len(d):6d} {sum(d) / len(d):9.2f} {min(d):8.2f} {max(d):8.2f} {g:12.2f}")

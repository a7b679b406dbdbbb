#!/usr/bin/env python
"""profiles/traffic.json from the PMC passes of tools/profile_gpu.sh: HBM bytes per launch of every fused-rollout
kernel = (FETCH_SIZE x 2 + WRITE_SIZE) KiB x 1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half of
a wide coalesced stream -- that is what the classic-control kernels' 16-byte-per-lane streams are; the Brax kernels
read 4 bytes per lane, an access width the guide calls uncalibrated: their figure is recorded with and without the
factor).  Usage: make_traffic_json.py <gpurun_out/prof_TAG> <source label>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, source = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {"carl::Pendulum": "pendulum", "carl::CartPole": "cartpole", "carl::AcrobotT<double>": "acrobot",
          "carl::MountainCarCont": "mountaincar_cont", "carl::MountainCar": "mountaincar"}


def per_kernel(sub, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
path = os.path.join(ROOT, "profiles", "traffic.json")
out = json.load(open(path))
recs = {}
for name in fetch:
    if "rollout_staged_kernel" in name:
        fam = next((v for k, v in FAMILY.items() if f"<{k}," in name), None)
        if fam:
            recs[fam] = dict(hbm_bytes_per_launch=int((2 * fetch[name] + write.get(name, 0.0)) * 1024),
                             fetch_size_kib_raw=round(fetch[name], 1), write_size_kib=round(write.get(name, 0.0), 1),
                             source=source)
    elif "brax_kernel<1" in name:  # one instantiation per model in a bench run: told apart by their size below
        recs.setdefault("_brax", []).append((name, fetch[name], write.get(name, 0.0)))
for fam, rec in recs.items():
    if fam != "_brax":
        out[f"{fam}:65536:250"] = rec
if "acrobot" in recs and "mountaincar" in recs:  # BASELINE config 3: the two launches of one mixed step
    a, m = recs["acrobot"], recs["mountaincar"]
    out["acrobot+mountaincar:65536:250"] = dict(hbm_bytes_per_launch=a["hbm_bytes_per_launch"] + m["hbm_bytes_per_launch"],
                                                source=source, note="sum of the two families' launches")
for name, f, w in recs.get("_brax", []):
    key = "brax:" + name.split("brax_kernel<")[1].split(">")[0].replace(" ", "")
    out[key] = dict(fetch_size_kib_raw=round(f, 1), write_size_kib=round(w, 1),
                    hbm_bytes_per_launch_fetch_x1=int((f + w) * 1024), hbm_bytes_per_launch_fetch_x2=int((2 * f + w) * 1024),
                    source=source, note="4-byte-per-lane reads: FETCH_SIZE factor uncalibrated (guide), both recorded")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_doc"}, indent=1))

#!/usr/bin/env python
"""profiles/kernel_times.json from the rocprofv3 kernel trace of the DRIVER'S command (tools/r04_final.sh): the kernel's
own average duration per workload, told apart by launch shape (workgroups) and duration class -- so that `frac_kernel`
in the bench line and the committed bench line (profiles/r04_bench_driver_cmd.json) come from ONE gpurun call on ONE
box (the pool's boxes differ by +-8 %).   Usage: kernel_times_from_driver_trace.py gpurun_out/r04_final"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = defaultdict(list)
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def avg(pattern, wgs, cls="~median"):
    hits = [(k, v) for k, v in rows.items() if pattern in k[0] and k[1] == wgs]
    assert len(hits) == 1, (pattern, wgs, [k for k, _ in hits])
    d = sorted(hits[0][1])
    med = d[len(d) // 2]
    sel = [x for x in d if (x < 0.5 * med if cls == "<med/2" else 0.5 * med <= x <= 1.5 * med)]  # (outliers -- first launches, probes -- left out)
    return sum(sel) / len(sel), len(sel), hits[0][0][0]


SRC = ("rocprofv3 --kernel-trace of the driver's command, dispatches told apart by launch shape (tools/r04_final.sh, "
       "tools/kernel_times_from_driver_trace.py; same gpurun call as profiles/r04_bench_driver_cmd.json)")
spec = {
    # (second template argument = the action storage kind: 0 int32, 2 uint8)
    "cartpole:65536:1000": [("rollout_staged_kernel<carl::CartPole, 0,", 256, "~median")],
    "cartpole:65536:250": [("rollout_staged_kernel<carl::CartPole, 0,", 256, "<med/2")],
    "cartpole:8192:1000": [("rollout_staged_kernel<carl::CartPole, 0,", 32, "~median")],
    "cartpole_narrow:65536:1000": [("rollout_staged_kernel<carl::CartPole, 2,", 256, "~median")],
    "pendulum:65536:1000": [("rollout_staged_kernel<carl::Pendulum, 0,", 256, "~median")],
    "pendulum_narrow:65536:1000": [("rollout_staged_kernel<carl::Pendulum, 3,", 256, "~median")],
    "pendulum:8192:1000": [("rollout_staged_kernel<carl::Pendulum, 0,", 32, "~median")],
    "acrobot+mountaincar:65536:1000": [("rollout_staged_pair_kernel<carl::AcrobotT<double>, carl::MountainCar", 512, "~median")],
    "acrobot+mountaincar:8192:1000": [("rollout_staged_pair_kernel<carl::AcrobotT<double>, carl::MountainCar", 64, "~median")],
    "ant:32768:20": [("brax_kernel<1, false, 9, false, false>", 256, "~median")],
    "halfcheetah+humanoid:32768:20": [("brax_kernel<1, true, 11, false, false>", 512, "~median"),
                                      ("brax_kernel<1, false, 7, false, true>", 768, "~median")],
}
path = os.path.join(ROOT, "profiles", "kernel_times.json")
out = json.load(open(path))
for key, parts in spec.items():
    tot, kern = 0.0, {}
    for pat, wgs, cls in parts:
        a, n, name = avg(pat, wgs, cls)
        tot += a
        kern[name[:110]] = dict(calls=n, avg_us=a, workgroups=wgs)
    out[key] = dict(kernel_avg_us=tot, source=SRC, kernels=kern)
    print(f"{key:34s} {tot:9.2f} us")
json.dump(out, open(path, "w"), indent=1)

#!/usr/bin/env python
"""profiles/kernel_times.json, profiles/brax_valu.json and the r04 entries of profiles/traffic.json from the passes of
tools/r04_evidence.sh, plus a text summary (profiles/r04_rocprofv3_summary.txt).

    python tools/make_r04_profiles.py gpurun_out/prof_r04 "<source label>" [r05]

Per workload directory <env>_<lanes>_<chunk>/: kt/ (kernel trace + stats), fetch/, write/ (PMC), sq/ (Brax).  Only the
ROLLOUT kernels count (rollout_staged_kernel / rollout_staged_pair_kernel / brax_kernel<1, ...>); a workload with two
rollout kernels per pass of the hot path (Halfcheetah + Humanoid: two launches) sums their per-dispatch averages.
HBM bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB x 1024 for the classic kernels (gfx950 reports half of a wide coalesced
read stream: MI355X_MICROARCH.md); the Brax kernels read 4 bytes per lane, a width the guide calls uncalibrated: both
factors are recorded."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, source = sys.argv[1], sys.argv[2]
ROUND = sys.argv[3] if len(sys.argv) > 3 else "r04"  # names the text summary: profiles/<round>_rocprofv3_summary.txt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRAX = ("ant", "halfcheetah", "humanoid")


def is_rollout(name):
    return "rollout_staged" in name or "brax_kernel<1" in name


def counters(wdir, sub):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(wdir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if is_rollout(r["Kernel_Name"]):
                a = acc[r["Kernel_Name"]][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}, {k: max(v[0] for v in d.values()) for k, d in acc.items()}


def kernel_stats(wdir):
    out = {}
    for f in glob.glob(os.path.join(wdir, "kt", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if is_rollout(r["Name"]):
                out[r["Name"]] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]),
                                      max_ns=float(r["MaxNs"]), pct=float(r["Percentage"]))
    return out


def load(path, default):
    try:
        return json.load(open(path))
    except OSError:
        return default


ktimes = load(os.path.join(ROOT, "profiles", "kernel_times.json"), {})
valu = load(os.path.join(ROOT, "profiles", "brax_valu.json"), {})
traffic = load(os.path.join(ROOT, "profiles", "traffic.json"), {})
lines = [f"rocprofv3 summaries of tools/r04_evidence.sh -- {source}", ""]
for wdir in sorted(glob.glob(os.path.join(root, "*_*_*"))):
    if not os.path.isdir(wdir):
        continue
    env, lanes, chunk = os.path.basename(wdir).rsplit("_", 2)
    key = f"{env}:{lanes}:{chunk}"
    ks = kernel_stats(wdir)
    lines.append(f"== {key}")
    if ks:
        ktimes[key] = dict(kernel_avg_us=sum(v["avg_ns"] for v in ks.values()) / 1e3, source=source,
                           kernels={k[:110]: dict(calls=v["calls"], avg_us=v["avg_ns"] / 1e3, min_us=v["min_ns"] / 1e3,
                                                  max_us=v["max_ns"] / 1e3) for k, v in ks.items()})
        for k, v in ks.items():
            lines.append(f"  kernel-trace  {k[:120]}\n                calls {v['calls']}  avg {v['avg_ns'] / 1e3:.2f} us  min {v['min_ns'] / 1e3:.2f}  "
                         f"max {v['max_ns'] / 1e3:.2f}  ({v['pct']:.1f} % of GPU time)")
    fetch, nf = counters(wdir, "fetch")
    write, _ = counters(wdir, "write")
    if fetch and write:
        f_kib = sum(d.get("FETCH_SIZE", 0.0) for d in fetch.values())
        w_kib = sum(d.get("WRITE_SIZE", 0.0) for d in write.values())
        brax = any(e in BRAX for e in env.split("+"))
        rec = dict(fetch_size_kib_raw=round(f_kib, 1), write_size_kib=round(w_kib, 1), source=source,
                   dispatches_averaged=sum(nf.values()))
        if brax:
            rec.update(hbm_bytes_per_launch=int((f_kib + w_kib) * 1024), hbm_bytes_per_launch_fetch_x2=int((2 * f_kib + w_kib) * 1024),
                       note="4-byte-per-lane reads: FETCH_SIZE factor uncalibrated (guide); x1 reported as the traffic, x2 beside it")
        else:
            rec.update(hbm_bytes_per_launch=int((2 * f_kib + w_kib) * 1024))
        traffic[key] = rec
        lines.append(f"  pmc           FETCH_SIZE {f_kib:.1f} KiB  WRITE_SIZE {w_kib:.1f} KiB per pass of the hot path -> {rec['hbm_bytes_per_launch'] / 1e6:.2f} MB")
    sq, nsq = counters(wdir, "sq")
    if sq:
        tot = defaultdict(float)
        for d in sq.values():
            for c, v in d.items():
                tot[c] += v
        valu[key] = dict(insts_valu_per_launch=tot["SQ_INSTS_VALU"], insts_salu_per_launch=tot["SQ_INSTS_SALU"],
                         insts_lds_per_launch=tot["SQ_INSTS_LDS"], waves_per_launch=tot["SQ_WAVES"], source=source,
                         dispatches_averaged=sum(nsq.values()),
                         per_kernel={k[:110]: d for k, d in sq.items()})
        lines.append("  pmc           " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(tot.items())) + " per pass of the hot path")
for name, obj in (("kernel_times", ktimes), ("brax_valu", valu), ("traffic", traffic)):
    json.dump(obj, open(os.path.join(ROOT, "profiles", name + ".json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{ROUND}_rocprofv3_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

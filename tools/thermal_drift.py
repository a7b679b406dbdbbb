#!/usr/bin/env python
"""Does the headline's launch time drift with the state of the card?  One CartPole x 65 536 x 1 000-step launch train,
run back to back for --seconds; every ~0.5 s one row: elapsed s, launch period us (HIP events over 100 launches),
sclk / mclk / fclk / socclk (sysfs), hwmon temperatures and power.  Then --idle seconds of nothing and a second, short
train (does it recover?).      python tools/thermal_drift.py [--seconds 45] [--idle 20]
(r04: inside one gpurun call the driver's command measured 274 us per launch first and 304 us a minute later.)"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--seconds", type=float, default=45.0)
    p.add_argument("--idle", type=float, default=20.0)
    p.add_argument("--env", default="cartpole")
    a = p.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.Workload(tuple(a.env.split("+")), 65536, 1000, 2, 0, 1, dev)
    clk = bench.ClockSampler(dev)

    def snapshot():
        row = {k: clk._read(k) for k in clk.samples}
        for k, (path, scale) in clk.hwmon.items():
            try:
                with open(path) as f:
                    row[k] = round(int(f.read()) * scale, 1)
            except (OSError, ValueError):
                pass
        return row

    def phase(name, seconds):
        t0 = time.time()
        first = True
        while time.time() - t0 < seconds:
            _, ev = wl.train(100, 5 if first else 0, lambda: None)
            first = False
            print(json.dumps({"phase": name, "t": round(time.time() - t0, 2), "launch_us": round(ev * 1e6, 1), **snapshot()}), flush=True)
            t_next = time.time() + 0.4
            while time.time() < t_next:  # keep the GPU busy between rows too
                wl.train(100, 0, lambda: None)

    print(json.dumps({"phase": "idle-before", **snapshot()}), flush=True)
    phase("busy", a.seconds)
    time.sleep(a.idle)
    print(json.dumps({"phase": "idle-after", "idle_s": a.idle, **snapshot()}), flush=True)
    phase("busy-again", 4.0)


if __name__ == "__main__":
    main()

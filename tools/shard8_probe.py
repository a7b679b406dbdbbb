#!/usr/bin/env python
"""Round-4 probe: the 8-GPU shard regime measured on ONE GPU, and the launch-boundary options of the headline.

    python tools/shard8_probe.py [--reps 5] [--only name,name]

Each record: median over `reps` trains of K launches (HIP events on the launch stream + wall clock).
Workloads are bench.py's own `Workload` objects (same context sets, same buffers, same launch train).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--only", default="")
    a = p.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cases = [
        # name, families, lanes per family, T, K, mode
        ("cartpole_65536_T250", ("cartpole",), 65536, 250, 20, "train"),
        ("cartpole_65536_T1000", ("cartpole",), 65536, 1000, 8, "train"),
        ("cartpole_u8_65536_T1000", ("cartpole",), 65536, 1000, 8, "train-u8"),
        ("cartpole_u8_65536_T250", ("cartpole",), 65536, 250, 20, "train-u8"),
        ("cartpole_u8_8192_T1000", ("cartpole",), 8192, 1000, 8, "train-u8"),
        ("mountaincar_u8_65536_T250", ("mountaincar",), 65536, 250, 20, "train-u8"),
        ("mountaincar_65536_T1000", ("mountaincar",), 65536, 1000, 8, "train"),
        ("mountaincar_u8_65536_T1000", ("mountaincar",), 65536, 1000, 8, "train-u8"),
        ("acrobot_65536_T1000", ("acrobot",), 65536, 1000, 8, "train"),
        ("acrobot_u8_65536_T1000", ("acrobot",), 65536, 1000, 8, "train-u8"),
        ("pendulum_f16_65536_T1000", ("pendulum",), 65536, 1000, 8, "train-u8"),
        ("pendulum_f16_65536_T250", ("pendulum",), 65536, 250, 20, "train-u8"),
        ("cartpole_2x32768_free_T250", ("cartpole", "cartpole"), 32768, 250, 20, "free"),
        ("cartpole_4x16384_free_T250", ("cartpole",) * 4, 16384, 250, 20, "free"),
        ("pendulum_65536_T250", ("pendulum",), 65536, 250, 20, "train"),
        ("pendulum_65536_T1000", ("pendulum",), 65536, 1000, 8, "train"),
        ("pendulum_2x32768_free_T250", ("pendulum", "pendulum"), 32768, 250, 20, "free"),
        ("cartpole_16384_T1000", ("cartpole",), 16384, 1000, 8, "train"),
        ("cartpole_32768_T1000", ("cartpole",), 32768, 1000, 8, "train"),
        ("pendulum_16384_T1000", ("pendulum",), 16384, 1000, 8, "train"),
        ("pendulum_32768_T1000", ("pendulum",), 32768, 1000, 8, "train"),
        ("cartpole_8192_T250", ("cartpole",), 8192, 250, 20, "train"),
        ("cartpole_8192_T1000", ("cartpole",), 8192, 1000, 8, "train"),
        ("pendulum_8192_T250", ("pendulum",), 8192, 250, 20, "train"),
        ("pendulum_8192_T1000", ("pendulum",), 8192, 1000, 8, "train"),
        ("config3_16384_T250", ("acrobot", "mountaincar"), 8192, 250, 20, "train"),
        ("config3_65536_T250", ("acrobot", "mountaincar"), 65536, 250, 20, "train"),
        ("config3_65536_T1000", ("acrobot", "mountaincar"), 65536, 1000, 8, "train"),
        ("config3_16384_T1000", ("acrobot", "mountaincar"), 8192, 1000, 8, "train"),
        ("acrobot_65536_T250", ("acrobot",), 65536, 250, 20, "train"),
        ("mountaincar_65536_T250", ("mountaincar",), 65536, 250, 20, "train"),
        ("ant_256_T20", ("ant",), 256, 20, 20, "train"),
        ("ant_1024_T20", ("ant",), 1024, 20, 20, "train"),
        ("ant_2048_T20", ("ant",), 2048, 20, 20, "train"),
        ("ant_4096_T20", ("ant",), 4096, 20, 20, "train"),
        ("ant_8192_T20", ("ant",), 8192, 20, 20, "train"),
        ("ant_32768_T20", ("ant",), 32768, 20, 10, "train"),
        ("humanoid_32768_T20", ("humanoid",), 32768, 20, 10, "train"),
        ("halfcheetah_32768_T20", ("halfcheetah",), 32768, 20, 10, "train"),
        ("cheetah_humanoid_4096_T20", ("halfcheetah", "humanoid"), 4096, 20, 20, "train"),
        ("halfcheetah_4096_T20", ("halfcheetah",), 4096, 20, 20, "train"),
        ("humanoid_4096_T20", ("humanoid",), 4096, 20, 20, "train"),
    ]
    only = set(a.only.split(",")) if a.only else None
    for name, fams, lanes, T, K, mode in cases:
        if only and name not in only:
            continue
        wl = bench.Workload(fams, lanes, T, 2, 0, 1, dev, narrow_actions=mode.endswith("-u8"))
        mode = mode.replace("-u8", "")
        per, walls = [], []
        for r in range(a.reps + 1):
            if mode == "free":
                w = wl.train_free_running(K, 5 if r == 0 else 0, lambda: None)
                ev = w / K
            else:
                w, ev = wl.train(K, 5 if r == 0 else 0, lambda: None)
            if r:  # first repetition = warm-up
                per.append(ev)
                walls.append(w / K)
        ev_med, wall_med = statistics.median(per), statistics.median(walls)
        rec = {"name": name, "lanes": wl.n, "T": T, "K": K, "event_us": ev_med * 1e6, "wall_us": wall_med * 1e6,
               "value_event": wl.n * T / ev_med, "value_wall": wl.n * T / wall_med,
               "frac_hbm_event": wl.bytes_per_launch / ev_med / 1e9 / bench.HBM_PEAK_GBS,
               "min_us": min(per) * 1e6, "max_us": max(per) * 1e6, "shape": wl.launch_shape()}
        print(json.dumps(rec), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

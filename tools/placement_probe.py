#!/usr/bin/env python
"""Is the headline's launch time a property of WHERE its buffers landed in VRAM?  In one process: build the CartPole x
65 536 x 1 000-step workload several times -- (a) freeing the previous one and emptying torch's cache first (fresh
hipMalloc), (b) keeping the previous ones alive (new addresses) -- and time 3 x 50 launches of each.  One row per build:
data pointers (obs buffer of set 0), launch period us of the three trains.
    python tools/placement_probe.py [--builds 6]"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--builds", type=int, default=6)
    p.add_argument("--skew", default="", help="comma-separated byte skews for the slab experiment")
    p.add_argument("--slabs", action="store_true")
    p.add_argument("--uniform-skew", action="store_true", help="--skew moves ALL arrays of a set by the same offset")
    p.add_argument("--act-skew", default="", help="comma-separated byte offsets of the action arrays past a 2 MiB boundary")
    a = p.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def measure(wl, tag, i):
        per = []
        for r in range(4):
            _, ev = wl.train(50, 5 if r == 0 else 0, lambda: None)
            if r:
                per.append(round(ev * 1e6, 1))
        ptrs = []
        for out_set in wl.outs:  # [set][part] -> the rollout's output record (obs first)
            o = out_set[0]
            t = o[0] if isinstance(o, (tuple, list)) else (next(iter(o.values())) if isinstance(o, dict) else getattr(o, "obs", o))
            ptrs.append(hex(t.data_ptr()) if hasattr(t, "data_ptr") else str(type(t)))
        print(json.dumps({"mode": tag, "build": i, "launch_us": per, "ptrs": ptrs,
                          "reserved_gb": round(torch.cuda.memory_reserved() / 2**30, 2)}), flush=True)

    if a.skew:
        # (c) the four output arrays of each buffer set carved out of ONE slab, array k shifted by k * skew bytes past its
        # 2 MiB-aligned slot: do the streams (obs 1 MiB / step, reward 256 KiB, flags 64 KiB each) collide when every
        # base is 2 MiB-aligned, as separate torch allocations are?
        wl = bench.Workload(("cartpole",), 65536, 1000, 2, 0, 1, dev)
        measure(wl, "separate-allocations", 0)
        MiB = 1 << 20
        shapes = {k: (tuple(v.shape), v.dtype) for k, v in wl.outs[0][0].items()}
        sizes = {k: v.numel() * v.element_size() for k, v in wl.outs[0][0].items()}
        slot = {k: (sz + 4 * MiB + 2 * MiB - 1) // (2 * MiB) * (2 * MiB) for k, sz in sizes.items()}
        slabs = [torch.empty(sum(slot.values()) + 2 * MiB, dtype=torch.uint8, device=dev) for _ in wl.outs]
        for skew in [int(x) for x in a.skew.split(",")]:
            for si, slab in enumerate(slabs):
                base = (-slab.data_ptr()) % (2 * MiB)
                off, carved = base, {}
                for k, name in enumerate(shapes):
                    shp, dt = shapes[name]
                    o = off + (skew if a.uniform_skew else k * skew)  # (uniform: every array, obs included, moves)
                    carved[name] = slab[o:o + sizes[name]].view(dt).view(shp)
                    off += slot[name]
                wl.outs[si][0] = carved
            measure(wl, f"slab-skew-{skew}", 0)
        return

    if a.act_skew:
        # (e) the ACTION arrays (the launch's only per-step read stream) moved relative to fixed output arrays: each
        # set's [T][N] int32 actions copied into a slab at `skew` bytes past a 2 MiB boundary
        wl = bench.Workload(("cartpole",), 65536, 1000, 2, 0, 1, dev)
        measure(wl, "actions-as-allocated", 0)
        MiB = 1 << 20
        src = [wl.acts[si][0] for si in range(len(wl.acts))]
        nbytes = src[0].numel() * src[0].element_size()
        slabs = [torch.empty(nbytes + 64 * MiB, dtype=torch.uint8, device=dev) for _ in src]
        for skew in [int(x) for x in a.act_skew.split(",")]:
            for si, slab in enumerate(slabs):
                off = (-slab.data_ptr()) % (2 * MiB) + skew
                view = slab[off:off + nbytes].view(src[si].dtype).view(src[si].shape)
                view.copy_(src[si])
                wl.acts[si][0] = view
            measure(wl, f"action-skew-{skew}", 0)
        return

    if a.slabs:
        # (d) separate torch allocations per array (what alloc_rollout does) against ONE slab per buffer set, alternating,
        # every build at new addresses (earlier ones stay alive): is one big allocation placed better than four?
        MiB = 1 << 20
        keep = []
        for i in range(a.builds):
            wl = bench.Workload(("cartpole",), 65536, 1000, 2, 0, 1, dev)
            measure(wl, "separate", i)
            keep.append([wl.outs[0][0], wl.outs[1][0]])
            sizes = {k: v.numel() * v.element_size() for k, v in wl.outs[0][0].items()}
            shapes = {k: (tuple(v.shape), v.dtype) for k, v in wl.outs[0][0].items()}
            slot = {k: (sz + 2 * MiB - 1) // (2 * MiB) * (2 * MiB) for k, sz in sizes.items()}
            for si in range(len(wl.outs)):
                slab = torch.empty(sum(slot.values()) + 2 * MiB, dtype=torch.uint8, device=dev)
                keep.append(slab)
                off, carved = (-slab.data_ptr()) % (2 * MiB), {}
                for name, (shp, dt) in shapes.items():
                    carved[name] = slab[off:off + sizes[name]].view(dt).view(shp)
                    off += slot[name]
                wl.outs[si][0] = carved
            measure(wl, "slab", i)
            keep.append(wl)
        return

    for i in range(a.builds):  # (a) one at a time, cache emptied in between
        wl = bench.Workload(("cartpole",), 65536, 1000, 2, 0, 1, dev)
        measure(wl, "fresh", i)
        del wl
        torch.cuda.empty_cache()
    keep = []
    for i in range(a.builds):  # (b) earlier builds stay alive: every build at new addresses
        wl = bench.Workload(("cartpole",), 65536, 1000, 2, 0, 1, dev)
        measure(wl, "stacked", i)
        keep.append(wl)
    for i, wl in enumerate(keep):  # and the same objects again, in order: is the time a property of the object?
        measure(wl, "stacked-again", i)


if __name__ == "__main__":
    main()

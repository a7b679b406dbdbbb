"""Round 4: where does the driver's 20-launch timed region lose its ~100 us against the sustained train?
Per-launch HIP-event intervals inside a region, after (a) an idle gap of 5 ms, (b) a minimal gap, (c) a gap filled with
a gc.collect() like bench.py's train().  Run on the GPU box."""
import gc
import sys
import time

sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
import torch

import bench

dev = torch.device("cuda", 0)
wl = bench.Workload(("cartpole",), 65536, 250, 2, 0, 1, dev)
for _ in range(200):
    wl.launch()
torch.cuda.synchronize()
K = 20
for label, gap in (("idle 5 ms", lambda: time.sleep(0.005)), ("minimal gap", lambda: None), ("gc.collect()", gc.collect),
                   ("idle 50 ms", lambda: time.sleep(0.05)), ("minimal gap", lambda: None)):
    for rep in range(3):
        for _ in range(5):
            wl.launch()
        torch.cuda.synchronize()
        gap()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for k in range(K):
            wl.launch()
            evs[k + 1].record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        iv = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(K)]
        print(f"{label:14s} rep {rep}: wall {1e6 * (t2 - t0):7.1f} us (enqueue {1e6 * (t1 - t0):6.1f}, sync tail {1e6 * (t2 - t1):7.1f}) "
              f"events total {sum(iv):7.1f} | first 4: {[round(x, 1) for x in iv[:4]]} median {sorted(iv)[K // 2]:.1f} last {iv[-1]:.1f}", flush=True)

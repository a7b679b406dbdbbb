"""Can two families' staged rollouts share the CUs?  acrobot + mountaincar, 65 536 lanes each: launches back to back on one
stream vs each family's train on its own stream, joined once at the end (no per-call fork / join).
    python tools/mixed_overlap.py        (on the GPU box; CARL_AMD_LIB_PATH selects the library)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

n, T = 65536, 250
wl = bench.Workload(("acrobot", "mountaincar"), n, T, 2, 0, 1, torch.device("cuda:0"))
eng = wl.eng
for mode in ("sequential", "own_streams", "sequential", "own_streams"):
    fr = mode == "own_streams"
    for i in range(6):
        eng.rollout(wl.acts[i % 2], wl.outs[i % 2], free_running=fr)
    eng.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 60
    for i in range(reps):
        eng.rollout(wl.acts[i % 2], wl.outs[i % 2], free_running=fr)
    eng.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"acrobot+mountaincar x {n} each, T={T}  {mode:12s} {dt * 1e6:8.1f} us per mixed launch  {2 * n * T / dt:.3e} env-steps/s", flush=True)

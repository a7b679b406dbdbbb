"""Practical HBM ceiling for the rollout's traffic pattern (run on the GPU box):
    python tools/ceiling/run_ceiling.py
Compiles stream_ceiling.hip (hipcc, gfx950), runs it with the bench's shapes (65 536 lanes, T = 250
and 1000) and prints time per launch and GB/s of the same 22 algorithmic bytes per lane-step."""
import ctypes as C
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libstream_ceiling.so")
flags = ["-DBLOCK_MAJOR"] if "--block-major" in sys.argv else []
flags += [a for a in sys.argv[1:] if a.startswith("-DLPB=")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-fno-gpu-rdc", *flags,
                os.path.join(here, "stream_ceiling.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.launch_stream.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
sink = torch.zeros(4, device="cuda")
n = 65536
for T in (250, 1000):
    act = torch.rand((T, n), device="cuda") * 4 - 2
    obs = torch.empty((T, n, 3), device="cuda")
    rew = torch.empty((T, n), device="cuda")
    term = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    trunc = torch.empty((T, n), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def go():
        rc = lib.launch_stream(act.data_ptr(), obs.data_ptr(), rew.data_ptr(), term.data_ptr(), trunc.data_ptr(), sink.data_ptr(), n, T, st)
        assert rc == 0, rc

    for _ in range(5):
        go()
    torch.cuda.synchronize()
    reps = 100 if T == 250 else 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"T={T}: {us:.1f} us per launch-to-launch, {us * 1e3 / T:.1f} ns/step, {22.0 * n * T / us / 1e3:.0f} GB/s "
          f"({22.0 * n * T / us / 1e3 / 8000:.3f} of 8 TB/s)")
    if not flags or flags == ["-DLPB=256"]:
        assert float(rew[T - 1, 5]) == float(T - 1 + 1) + 1.0  # row T-1, lane 5 = wave lane 1, component 1
sys.stdout.flush()

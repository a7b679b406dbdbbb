"""Round 4: which stream of the rollout's traffic pattern costs the HBM rate?  (run on the GPU box)
    python tools/ceiling/run_pattern_probe.py
Times tools/ceiling/pattern_probe.hip's modes at CartPole's shapes (65 536 lanes, 250-step launches, two rotating
buffer sets > the Infinity Cache) and prints bytes moved / time."""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libpattern_probe.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-fno-gpu-rdc",
                os.path.join(here, "pattern_probe.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.launch_probe.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_void_p]
n, T = 65536, 250
sink = torch.zeros(4, device="cuda")
sets = []
for _ in range(2):
    sets.append((torch.randint(0, 2, (T, n), device="cuda", dtype=torch.int32), torch.empty((T, n, 4), device="cuda"),
                 torch.empty((T, n), device="cuda"), torch.empty((T, n), dtype=torch.uint8, device="cuda"),
                 torch.empty((T, n), dtype=torch.uint8, device="cuda")))
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "all five streams (read 4 + write 16 + 4 + 1 + 1 B)", 1: "no reader (write 22 B)", 2: "no flag rows (read 4 + write 20 B)",
         3: "obs only (write 16 B)", 4: "reader + obs (read 4 + write 16 B)", 5: "all five, temporal stores", 6: "plain fill of the obs buffer (16 B x lanes x T)", 7: "reader alone, 8 rows in flight",
         8: "reader alone, pipelined (8 + 8 in flight)", 9: "all five streams, pipelined reader", 10: "all five, reads in bursts of 32 rows",
         11: "all five, reads in bursts of 64 rows", 12: "all five, reads in bursts of 128 rows", 13: "all five, loads without cache bits",
         14: "all five, loads sc0", 15: "all five, loads sc1", 16: "all five, loads sc0 sc1", 17: "all five, every workgroup reads workgroup 0's piece (L2 hits)",
         18: "all five, the same 8 rows again and again (no HBM reads)",
         19: "all five, each workgroup re-reads its own 8 KiB (cacheable)", 20: "all five, each workgroup re-reads its own 8 KiB (nt)"}
BYTES = {0: 26, 1: 22, 2: 24, 3: 16, 4: 20, 5: 26, 6: 16, 7: 4, 8: 4, 9: 26, 10: 26, 11: 26, 12: 26, 13: 26, 14: 26, 15: 26, 16: 26, 17: 26, 18: 26, 19: 26, 20: 26}
for rep in range(2):
    for mode in (0, 1, 19, 20):
        k = [0]

        def go():
            a, o, r, te, tr = sets[k[0] % 2]
            k[0] += 1
            assert lib.launch_probe(a.data_ptr(), o.data_ptr(), r.data_ptr(), te.data_ptr(), tr.data_ptr(), sink.data_ptr(), n, T, mode, st) == 0

        for _ in range(10):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        gb = BYTES[mode] * n * T / us / 1e3
        print(f"rep {rep} mode {mode} {NAMES[mode]:52s}: {us:6.1f} us per launch  {gb:6.0f} GB/s  ({gb / 8000:.3f} of 8 TB/s)", flush=True)

#!/usr/bin/env python
"""Measured HBM ceilings on this box with plain torch kernels: fill (write-only), copy
(read+write), sum (read-only).  Context for roofline fractions of write-heavy kernels."""
import torch

dev = torch.device("cuda")
n = 1 << 28  # 1 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


b = n * 4
print(f"fill  (write-only): {b / timeit(lambda: x.fill_(1.0)) / 1e12:.2f} TB/s")
print(f"copy  (read+write): {2 * b / timeit(lambda: y.copy_(x)) / 1e12:.2f} TB/s")
print(f"sum   (read-only) : {b / timeit(lambda: x.sum()) / 1e12:.2f} TB/s")
xb = torch.empty(n, dtype=torch.uint8, device=dev)
print(f"fill u8 (write-only, 1 B/elem): {n / timeit(lambda: xb.fill_(1)) / 1e12:.2f} TB/s")

"""Fused CartPole rollout under the three device selector rules (static / round robin -- the reference's default --
/ random), with a context set that fits LDS (1 000) and one that does not (65 536).  Run on the GPU box."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from tests.test_gpu_parity import _engine, random_table, random_actions
from oracle import oracle as O
n, T = 65536, 250
rng = np.random.default_rng(0)
for C in (1000, 65536):
    table = random_table(O.CARTPOLE, rng, C)
    for sel, name in ((O.SEL_STATIC, "static"), (O.SEL_ROUND_ROBIN, "round_robin"), (O.SEL_RANDOM, "random")):
        e = _engine(O.CARTPOLE, table, n, "cuda", selector=sel, seed=1)
        e.reset()
        a = [torch.randint(0, 2, (T, n), device="cuda", dtype=torch.int32) for _ in range(2)]
        outs = [e.alloc_rollout(T) for _ in range(2)]
        for i in range(10): e.rollout(a[i % 2], outs[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): e.rollout(a[i % 2], outs[i % 2])
        e1.record(); torch.cuda.synchronize()
        print(f"CartPole 65536 lanes, {C} contexts, {name:12s}: {e0.elapsed_time(e1) / 100 * 1e3 / T * 1e3:.1f} ns/step", flush=True)

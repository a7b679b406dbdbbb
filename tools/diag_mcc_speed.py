"""Is the fused rollout's speed data-dependent?  MountainCarContinuous with random vs constant actions (the reward
stream becomes a constant), MountainCar with random vs constant actions."""
import sys, time; sys.path.insert(0, ".")
import torch
from carl_amd.context.selection import StaticSelector
from carl_amd.envs import CARLMountainCar, CARLMountainCarContinuous, CARLPendulum
n, T = 65536, 250
for cls, name in ((CARLMountainCarContinuous, "mcc"), (CARLMountainCar, "mc"), (CARLPendulum, "pendulum")):
    env = cls(num_envs=n, device="cuda:0", context_selector=StaticSelector); eng = env.env; env.reset(seed=0)
    info = eng.info
    for mode in ("random", "constant"):
        if info.action_is_discrete:
            a = [torch.randint(0, 3, (T, n), device="cuda", dtype=torch.int32) if mode == "random" else torch.ones((T, n), device="cuda", dtype=torch.int32) for _ in range(2)]
        else:
            lo, hi = float(info.action_low), float(info.action_high)
            a = [torch.rand((T, n), device="cuda") * (hi - lo) + lo if mode == "random" else torch.zeros((T, n), device="cuda") for _ in range(2)]
        outs = [eng.alloc_rollout(T) for _ in range(2)]
        for i in range(20): eng.rollout(a[i % 2], outs[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): eng.rollout(a[i % 2], outs[i % 2])
        e1.record(); torch.cuda.synchronize()
        print(f"{name:9s} {mode:9s} {e0.elapsed_time(e1) / 200 * 1e3 / T * 1e3:.1f} ns/step")

"""Humanoid / HumanoidStandup / Halfcheetah under BASELINE config 5's context distribution (joint_stiffness x U(0.5, 2), gravity
U(-15, -5)): N envs x T steps of full-range random actions, every observation finite and bounded.   (GPU box)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from carl_amd import envs as E  # noqa: E402
from carl_amd.context.context_space import UniformFloatContextFeature as U  # noqa: E402
from carl_amd.context.sampler import ContextSampler  # noqa: E402
from carl_amd.context.selection import StaticSelector  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for cls in (E.CARLBraxHumanoidStiffness, E.CARLBraxHalfcheetahStiffness):
    table = ContextSampler([U("joint_stiffness", 0.5, 2.0), U("gravity", -15, -5)], cls.get_context_space(), seed=0).sample_context_table(n)
    env = cls(contexts=table, batch_size=n, device="cuda:0", context_selector=StaticSelector, seed=0)
    eng = env.env
    env.reset(seed=0)
    lo, hi = float(min(eng.sys.act_lo[: eng.sys.n_act])), float(max(eng.sys.act_hi[: eng.sys.n_act]))
    chunk = 50
    out = eng.alloc_rollout(chunk)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    bad = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    worst = 0.0
    for k in range(T // chunk):
        a = torch.rand((chunk, n, eng.sys.n_act), device="cuda:0", generator=g) * (hi - lo) + lo
        eng.rollout(a, out)
        o = out["obs"]
        bad |= ((~torch.isfinite(o)).any(dim=2) | (o.abs().amax(dim=2) > 1000)).any(dim=0)
        worst = max(worst, float(torch.nan_to_num(o.abs(), nan=1e30).max()))
    print(f"{cls.__name__:32s} {n} envs x {T} steps: envs ever non-finite or |obs| > 1000: {int(bad.sum())}, max |obs| {worst:.1f}, "
          f"episodes {int(eng.episodes_done.sum())}", flush=True)

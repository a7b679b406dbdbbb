"""Long-run soak of the Brax kernels on the GPU box: N envs x T steps of random actions in fused
launches, every observation / reward finite, episode bookkeeping consistent.
    python tools/soak_brax.py [n_envs] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from carl_amd import envs as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chunk = 50
for cls in (E.CARLBraxAnt, E.CARLBraxHalfcheetah, E.CARLBraxHumanoid, E.CARLBraxHopper, E.CARLBraxWalker2d,
            E.CARLBraxInvertedPendulum, E.CARLBraxHumanoidStandup, E.CARLBraxInvertedDoublePendulum, E.CARLBraxReacher, E.CARLBraxPusher):
    env = cls(batch_size=n, device="cuda:0")
    eng = env.env
    env.reset(seed=0)
    lo, hi = float(min(eng.sys.act_lo[: eng.sys.n_act])), float(max(eng.sys.act_hi[: eng.sys.n_act]))
    out = eng.alloc_rollout(chunk)
    t0 = time.perf_counter()
    n_done = 0
    worst = 0.0
    for k in range(T // chunk):
        a = torch.rand((chunk, n, eng.sys.n_act), device="cuda:0") * (hi - lo) + lo
        eng.rollout(a, out)
        assert torch.isfinite(out["obs"]).all() and torch.isfinite(out["reward"]).all(), (cls.__name__, k)
        worst = max(worst, float(out["obs"].abs().max()))
        n_done += int((out["terminated"] | out["truncated"]).sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert n_done == int(eng.episodes_done.sum())
    print(f"{cls.__name__:28s} {n} envs x {T} steps ok: {n_done} episodes, max |obs| {worst:.1f}, "
          f"{n * T / dt:.3e} env-steps/s incl. action generation and checks, width {eng.sys.lanes_per_env}")

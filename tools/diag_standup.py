"""Where does HumanoidStandup's largest observation entry come from (round 6: max |obs| 7.6e3 in the soak after the
spring-backend gear override, 1.2e2 before)?  python tools/diag_standup.py [n_envs] [steps]   (GPU box)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from carl_amd import envs as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for label, scale in (("gears 350/100 (spring branch)", None), ("MJCF gears", [100, 100, 100, 100, 100, 300, 200, 100, 100, 300, 200, 25, 25, 25, 25, 25, 25])):
    env = E.CARLBraxHumanoidStandup(batch_size=n, device="cuda:0")
    eng = env.env
    if scale is not None:
        for k, g in enumerate(scale):
            eng.sys.act_gear[k] = float(g)
        eng.sys_dev = torch.frombuffer(bytearray(bytes(eng.sys)), dtype=torch.uint8).to(eng.device)
    env.reset(seed=0)
    chunk = 50
    out = eng.alloc_rollout(chunk)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    worst = torch.zeros(eng.sys.obs_dim, device="cuda:0")
    when = {}
    for k in range(T // chunk):
        a = torch.rand((chunk, n, eng.sys.n_act), device="cuda:0", generator=g) * 0.8 - 0.4
        eng.rollout(a, out)
        m = out["obs"].abs().amax(dim=(0, 1))
        worst = torch.maximum(worst, m)
        big = (out["obs"].abs().amax(dim=2) > 500).nonzero()
        if len(big) and "first" not in when:
            t, e = int(big[0, 0]), int(big[0, 1])
            when["first"] = (k * chunk + t, e)
            o = out["obs"][t, e].cpu().numpy()
            idx = np.argsort(-np.abs(o))[:8]
            print(label, "first |obs| > 500 at step", k * chunk + t, "env", e, [(int(i), float(o[i])) for i in idx])
            z = out["obs"][: t + 1, e, 0].cpu().numpy()
            print("   torso z over the chunk:", np.round(z[max(0, t - 10): t + 1], 3))
            print("   envs above 500 in this chunk:", int((out["obs"].abs().amax(dim=(0, 2)) > 500).sum()))
    top = torch.argsort(-worst)[:8].cpu().numpy()
    print(label, "max per entry:", [(int(i), float(worst[i])) for i in top])
